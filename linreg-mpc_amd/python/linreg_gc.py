"""ctypes binding of the product C ABI (include/linreg_gc.h, liblinreg_gc.so).

This is plumbing above the C ABI; all compute happens in the HIP library.
There is no CPU fallback: `Solver` raises `LgcError` (LGC_ENODEVICE) when no
MI355X is visible, and importing fails loudly when the library is not built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LGC_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "liblinreg_gc.so")

ALG = {"cholesky": 0, "ldlt": 1, "cgd": 2, "dimcheck": 3}   # dimcheck: the two parties' dimension comparison (src/linear.oc:109-114), not a solver


class LgcError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "lgc error %d: %s" % (code, msg))
        self.code = code


class System(C.Structure):
    """counterpart of linear_system_t (reference src/linear.h:16-26)"""
    _fields_ = [("d", C.c_size_t), ("width", C.c_int), ("precision", C.c_int), ("algorithm", C.c_int),
                ("num_iterations", C.c_int), ("lam", C.c_double), ("nshares", C.c_size_t),
                ("normalize", C.c_int), ("reveal_inputs", C.c_int), ("trace", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("and_gates", C.c_uint64), ("gate_steps", C.c_uint64), ("table_bytes", C.c_uint64),
                ("launches", C.c_uint64), ("seconds_total", C.c_double), ("seconds_garble", C.c_double),
                ("seconds_eval", C.c_double), ("seconds_mac_garble", C.c_double), ("seconds_mac_eval", C.c_double),
                ("mac_gates", C.c_uint64), ("mac_launches", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Record(C.Structure):
    _fields_ = [("op", C.c_uint32), ("cnt", C.c_uint32), ("dst", C.c_uint32), ("a", C.c_uint32),
                ("b", C.c_uint32), ("c", C.c_uint32), ("sa", C.c_int32), ("sb", C.c_int32), ("step0", C.c_uint64)]


class Launch(C.Structure):
    _fields_ = [("first_rec", C.c_uint32), ("nrec", C.c_uint32), ("step0", C.c_uint64), ("steps", C.c_uint64),
                ("gates", C.c_uint64), ("mac_only", C.c_int)]


class ProgramInfo(C.Structure):
    _fields_ = [("n_records", C.c_size_t), ("n_launches", C.c_size_t), ("n_words", C.c_uint32),
                ("n_reveal", C.c_uint32), ("in_base", C.c_uint32), ("rv_beta", C.c_uint32),
                ("rv_trace", C.c_uint32), ("rv_inputs", C.c_uint32), ("total_steps", C.c_uint64),
                ("total_gates", C.c_uint64), ("max_launch_steps", C.c_uint64),
                ("replicas", C.c_uint32), ("word_stride", C.c_uint32), ("reveal_stride", C.c_uint32),
                ("shared_end", C.c_uint32), ("prefix_launches", C.c_uint32), ("prefix_steps", C.c_uint64),
                ("total_xors", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C linreg-mpc_amd/csrc`)" % LIB_PATH)
        # A process that also uses torch on the GPU must import torch BEFORE this library is loaded: torch ships
        # its own libamdhip64 under the same SONAME and whichever loads first serves both (bench.py, tests/conftest.py)
        L = C.CDLL(LIB_PATH)
        vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
        L.lgc_last_error.restype = C.c_char_p
        L.lgc_version.restype = C.c_char_p
        L.lgc_device_count.restype = ci
        for name, args in [
            ("lgc_solver_create", [C.POINTER(vp), ci, C.POINTER(System), C.c_char_p]),
            ("lgc_solver_set_shares", [vp, vp]), ("lgc_solver_run", [vp, ci]),
            ("lgc_solver_get_beta", [vp, vp]), ("lgc_solver_get_trace", [vp, vp]),
            ("lgc_solver_get_inputs", [vp, vp]), ("lgc_solver_get_stats", [vp, C.POINTER(Stats)]),
            ("lgc_solver_get_profile", [vp, vp, vp, sz]), ("lgc_solver_get_iterations", [vp, vp, vp, sz]),
            ("lgc_party_iteration_marks", [vp, vp, vp, sz]),
            ("lgc_program_build", [C.POINTER(vp), C.POINTER(System)]),
            ("lgc_program_build_sweep", [C.POINTER(vp), C.POINTER(System), sz, vp]),
            ("lgc_program_build_sweep_at", [C.POINTER(vp), C.POINTER(System), sz, vp, sz]),
            ("lgc_solver_create_sweep", [C.POINTER(vp), ci, C.POINTER(System), C.c_char_p, sz, vp]),
            ("lgc_solver_create_sweep_at", [C.POINTER(vp), ci, C.POINTER(System), C.c_char_p, sz, vp, sz]),
            ("lgc_solver_prefix_garble", [vp]), ("lgc_solver_prefix_export", [vp, vp]), ("lgc_solver_prefix_import", [vp, vp]),
            ("lgc_program_info_get", [vp, C.POINTER(ProgramInfo)]),
            ("lgc_program_ring_plan", [vp, sz, C.POINTER(sz), vp, vp]),
            ("lgc_aes_bench", [ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
            ("lgc_p1_create", [C.POINTER(vp), ci, sz, sz, ci, ci]), ("lgc_p1_set_data", [vp, vp, vp]),
            ("lgc_p1_local", [vp, sz, sz, ci, vp, vp]), ("lgc_p1_mask", [vp, vp, sz, vp, ci, vp]),
            ("lgc_p1_dot", [vp, vp, vp, vp, sz, vp, vp]),
            ("lgc_p1_ti_a", [vp, C.c_uint32, vp, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]),
            ("lgc_ti_generate", [ci, C.c_char_p, C.c_uint64, sz, sz, ci, vp, vp, vp, vp]),
            ("lgc_party_create", [C.POINTER(vp), ci, C.POINTER(System), ci, C.c_char_p, sz]),
            ("lgc_party_input_pairs", [vp, sz, vp, vp]), ("lgc_party_encode_inputs", [vp, sz, vp, vp]),
            ("lgc_party_set_input_labels", [vp, sz, vp]), ("lgc_party_garble", [vp, sz, vp]),
            ("lgc_party_evaluate", [vp, sz, vp]), ("lgc_party_decode_bits", [vp, vp]),
            ("lgc_party_finish", [vp, vp, vp, vp, vp]),
            ("lgc_party_ring_create", [vp, ci, vp, C.POINTER(sz)]), ("lgc_party_ring_open", [vp, vp, ci, sz]),
            ("lgc_party_garble_ring", [vp, sz]), ("lgc_party_evaluate_ring", [vp, sz]),
            ("lgc_party_garble_ring_begin", [vp, sz]), ("lgc_party_garble_ring_wait", [vp, sz]),
            ("lgc_party_garble_ring_streams", [vp, C.c_int]),
            ("lgc_test_party_garble_ring_stage", [vp, sz, ci, C.POINTER(ci)]), ("lgc_test_party_ring_read", [vp, sz, vp, sz]),
            ("lgc_ot_sender_create", [C.POINTER(vp), ci, C.c_char_p, vp]),
            ("lgc_ot_receiver_create", [C.POINTER(vp), ci, vp, vp]),
            ("lgc_ot_sender_set_device_io", [vp, ci]), ("lgc_ot_receiver_set_device_io", [vp, ci]),
            ("lgc_ot_gilboa_recv_start", [vp, vp, sz, sz, ci, vp]),
            ("lgc_ot_gilboa_send", [vp, vp, sz, sz, ci, vp, vp, vp]),
            ("lgc_ot_gilboa_recv_finish", [vp, vp, vp]),
            ("lgc_ot_labels_recv_start", [vp, vp, sz, vp]), ("lgc_ot_labels_send", [vp, vp, vp, sz, vp, vp]),
            ("lgc_ot_labels_recv_finish", [vp, vp, vp]),
            ("lgc_aes_encrypt", [ci, vp, vp, sz]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = ci, args
        L.lgc_set_split_kernels.argtypes = [ci, ci]; L.lgc_set_split_kernels.restype = None
        L.lgc_release_cached_memory.argtypes = []; L.lgc_release_cached_memory.restype = None
        L.lgc_solver_destroy.argtypes = [vp]; L.lgc_solver_destroy.restype = None
        L.lgc_solver_prefix_bytes.argtypes = [vp]; L.lgc_solver_prefix_bytes.restype = sz
        L.lgc_program_destroy.argtypes = [vp]; L.lgc_program_destroy.restype = None
        L.lgc_p1_destroy.argtypes = [vp]; L.lgc_p1_destroy.restype = None
        L.lgc_party_destroy.argtypes = [vp]; L.lgc_party_destroy.restype = None
        for nme in ("lgc_party_num_launches", "lgc_party_input_bits", "lgc_party_num_reveal"):
            getattr(L, nme).argtypes = [vp]; getattr(L, nme).restype = sz
        L.lgc_party_table_bytes.argtypes = [vp, sz]; L.lgc_party_table_bytes.restype = sz
        L.lgc_party_and_gates.argtypes = [vp]; L.lgc_party_and_gates.restype = C.c_uint64
        L.lgc_ot_sender_destroy.argtypes = [vp]; L.lgc_ot_sender_destroy.restype = None
        L.lgc_ot_receiver_destroy.argtypes = [vp]; L.lgc_ot_receiver_destroy.restype = None
        L.lgc_ot_u_bytes.argtypes = [C.c_uint64]; L.lgc_ot_u_bytes.restype = sz
        L.lgc_program_records.argtypes = [vp]; L.lgc_program_records.restype = C.POINTER(Record)
        L.lgc_program_launches.argtypes = [vp]; L.lgc_program_launches.restype = C.POINTER(Launch)
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise LgcError(rc, lib().lgc_last_error().decode())


def host_alloc(nbytes):
    """page-locked host buffer as a numpy uint8 array (lgc_host_alloc); free with host_free(arr)"""
    L = lib()
    L.lgc_host_alloc.restype = C.c_void_p; L.lgc_host_alloc.argtypes = [C.c_size_t]
    p = L.lgc_host_alloc(nbytes)
    if not p:
        raise LgcError(-4, L.lgc_last_error().decode())
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,))
    return arr


def host_free(arr):
    L = lib()
    L.lgc_host_free.restype = None; L.lgc_host_free.argtypes = [C.c_void_p]
    L.lgc_host_free(C.c_void_p(arr.ctypes.data))


def device_count():
    return lib().lgc_device_count()


def make_system(d, width=64, precision=56, algorithm="cgd", num_iterations=0, lam=0.0, nshares=2,
                normalize=0, reveal_inputs=0, trace=0):
    alg = ALG[algorithm] if isinstance(algorithm, str) else int(algorithm)
    return System(d, width, precision, alg, num_iterations, lam, nshares, normalize, reveal_inputs, trace)


class Program:
    """The lowered circuit program (host only; needs no GPU)."""

    def __init__(self, system, lambdas=None, first=0):
        self._h = C.c_void_p()
        if lambdas is None:
            _chk(lib().lgc_program_build(C.byref(self._h), C.byref(system)))
        else:                                    # per-lambda sweep: len(lambdas) circuits in one program
            lam = np.ascontiguousarray(lambdas, dtype=np.float64)   # (circuits first .. of a larger sweep)
            _chk(lib().lgc_program_build_sweep_at(C.byref(self._h), C.byref(system), lam.size, lam.ctypes.data_as(C.c_void_p), first))
        self.info = ProgramInfo()
        _chk(lib().lgc_program_info_get(self._h, C.byref(self.info)))
        self.system = system

    def records(self):
        n = self.info.n_records
        ptr = lib().lgc_program_records(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * C.sizeof(Record),)).copy()

    def launches(self):
        ptr = lib().lgc_program_launches(self._h)
        return [dict(first_rec=ptr[i].first_rec, nrec=ptr[i].nrec, step0=ptr[i].step0, steps=ptr[i].steps,
                     gates=ptr[i].gates, mac_only=ptr[i].mac_only) for i in range(self.info.n_launches)]

    def ring_plan(self, ring_bytes=0):
        """(ring size, offsets, wait_for) of the co-located solver's garbled-table ring"""
        n = self.info.n_launches
        off = np.zeros(n, dtype=np.uint64); wait = np.zeros(n, dtype=np.int64); rb = C.c_size_t()
        _chk(lib().lgc_program_ring_plan(self._h, ring_bytes, C.byref(rb), off.ctypes.data_as(C.c_void_p),
                                         wait.ctypes.data_as(C.c_void_p)))
        return rb.value, off, wait

    def close(self):
        if self._h:
            lib().lgc_program_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Solver:
    """Garble + evaluate one linear system on one MI355X (both roles co-located).

    Replaces `execYaoProtocol(pd, solver, &ls)` (reference src/cmd/linreg.c:177)."""

    def __init__(self, system, seed=b"\x01" * 16, device=0, lambdas=None, first=0):
        """lambdas: per-lambda sweep -- len(lambdas) circuits on the same shares in one program
        (lgc_solver_create_sweep); beta() then returns (len(lambdas), d).  first: index of lambdas[0]
        in a sweep sharded over several GPUs (lgc_solver_create_sweep_at; all ranks share the seed)."""
        assert len(seed) == 16
        self._h = C.c_void_p()
        self.system = system
        self.count = None
        if lambdas is None:
            _chk(lib().lgc_solver_create(C.byref(self._h), device, C.byref(system), seed))
        else:
            lam = np.ascontiguousarray(lambdas, dtype=np.float64)
            self.count = int(lam.size)
            _chk(lib().lgc_solver_create_sweep_at(C.byref(self._h), device, C.byref(system), seed, lam.size,
                                                  lam.ctypes.data_as(C.c_void_p), first))

    # ---- shared prefix of a sweep block (input labels + garbled share summation): garbled on one rank,
    # broadcast, imported by every rank.  dev_ptr: device memory of prefix_bytes() bytes (e.g. tensor.data_ptr())
    def prefix_bytes(self):
        return int(lib().lgc_solver_prefix_bytes(self._h))

    def prefix_garble(self):
        _chk(lib().lgc_solver_prefix_garble(self._h))

    def prefix_export(self, dev_ptr):
        _chk(lib().lgc_solver_prefix_export(self._h, C.c_void_p(dev_ptr)))

    def prefix_import(self, dev_ptr):
        _chk(lib().lgc_solver_prefix_import(self._h, C.c_void_p(dev_ptr)))

    def set_shares(self, shares):
        d = self.system.d
        shares = np.ascontiguousarray(shares, dtype=np.uint64)
        assert shares.size == self.system.nshares * (d * (d + 1) // 2 + d), shares.shape
        _chk(lib().lgc_solver_set_shares(self._h, shares.ctypes.data_as(C.c_void_p)))

    def run(self, profile=False):
        _chk(lib().lgc_solver_run(self._h, 1 if profile else 0))

    def beta(self):
        out = np.zeros(self.system.d if self.count is None else (self.count, self.system.d), dtype=np.int64)
        _chk(lib().lgc_solver_get_beta(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def trace(self):
        out = np.zeros((self.system.num_iterations, self.system.d + 4), dtype=np.int64)
        _chk(lib().lgc_solver_get_trace(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def inputs(self):
        d = self.system.d
        out = np.zeros(d * (d + 1) // 2 + d, dtype=np.int64)
        _chk(lib().lgc_solver_get_inputs(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self):
        st = Stats()
        _chk(lib().lgc_solver_get_stats(self._h, C.byref(st)))
        return st.asdict()

    def profile(self, nlaunches):
        g = np.zeros(nlaunches); e = np.zeros(nlaunches)
        _chk(lib().lgc_solver_get_profile(self._h, g.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), nlaunches))
        return g, e

    def iterations(self):
        """cgd: (cumulative AND gates, device seconds since the start of run) per iteration -- the
        values src/cgd.oc:190-194 prints as 'Iteration t gate count' / 'Iteration t time'."""
        n = int(self.system.num_iterations) if int(self.system.algorithm) == ALG["cgd"] else 0
        g = np.zeros(n, dtype=np.uint64); t = np.zeros(n)
        _chk(lib().lgc_solver_get_iterations(self._h, g.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), n))
        return g, t

    def close(self):
        if self._h:
            lib().lgc_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Phase1:
    """One data provider's quantised data on the device (phase-1 aggregation arithmetic,
    reference src/phase1.c)."""

    def __init__(self, Xq, yq=None, width=64, precision=56, device=0):
        Xq = np.ascontiguousarray(Xq, dtype=np.int64)
        self.n, self.d = Xq.shape
        self.w, self.p = width, precision
        self._h = C.c_void_p()
        _chk(lib().lgc_p1_create(C.byref(self._h), device, self.n, self.d, width, precision))
        yq = None if yq is None else np.ascontiguousarray(yq, dtype=np.int64)
        _chk(lib().lgc_p1_set_data(self._h, _vp(Xq), _vp(yq)))

    def local(self, c0, c1, with_y=False):
        own = c1 - c0
        A = np.zeros(own * (own + 1) // 2, dtype=np.uint64)
        b = np.zeros(own, dtype=np.uint64)
        _chk(lib().lgc_p1_local(self._h, c0, c1, 1 if with_y else 0, _vp(A), _vp(b)))
        return (A, b) if with_y else A

    def mask(self, cols, V, sign):
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        V = np.ascontiguousarray(V, dtype=np.uint64).reshape(len(cols), self.n)
        out = np.empty_like(V)
        _chk(lib().lgc_p1_mask(self._h, _vp(cols), len(cols), _vp(V), sign, _vp(out)))
        return out

    def ti_a(self, col, y, inn, sub):
        """party a of one inner_product_ti in a single pass: (a - y, <inn, y> - sub)"""
        y = np.ascontiguousarray(y, dtype=np.uint64).reshape(self.n)
        inn = np.ascontiguousarray(inn, dtype=np.uint64).reshape(self.n)
        out = np.empty(self.n, dtype=np.uint64); share = C.c_uint64()
        _chk(lib().lgc_p1_ti_a(self._h, int(col), _vp(y), _vp(inn), C.c_uint64(int(sub)), _vp(out), C.byref(share)))
        return out, np.uint64(share.value)

    def dot(self, A, B=None, cols=None, sub=None):
        A = np.ascontiguousarray(A, dtype=np.uint64).reshape(-1, self.n)
        npairs = A.shape[0]
        B = None if B is None else np.ascontiguousarray(B, dtype=np.uint64).reshape(npairs, self.n)
        cols = None if cols is None else np.ascontiguousarray(cols, dtype=np.uint32)
        sub = None if sub is None else np.ascontiguousarray(sub, dtype=np.uint64)
        out = np.zeros(npairs, dtype=np.uint64)
        _chk(lib().lgc_p1_dot(self._h, _vp(A), _vp(B), _vp(cols), npairs, _vp(sub), _vp(out)))
        return out

    def close(self):
        if self._h:
            lib().lgc_p1_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ti_generate(seed, first_pair, npairs, n, width=64, device=0):
    """(x, y, r, xy_minus_r) of the trusted initializer for `npairs` cross-party pairs"""
    x = np.zeros((npairs, n), dtype=np.uint64); y = np.zeros((npairs, n), dtype=np.uint64)
    r = np.zeros(npairs, dtype=np.uint64); xyr = np.zeros(npairs, dtype=np.uint64)
    _chk(lib().lgc_ti_generate(device, seed, first_pair, npairs, n, width, _vp(x), _vp(y), _vp(r), _vp(xyr)))
    return x, y, r, xyr


GARBLER, EVALUATOR = 1, 2


class Party:
    """CSP (garbler, role 1) or Evaluator (role 2) on its own: the host carries tables, labels
    and decode bits between the two (reference src/cmd/linreg.c:145-199, src/input.c)."""

    def __init__(self, system, role, seed=None, device=0, max_launch_table_bytes=0):
        self._h = C.c_void_p()
        self.system, self.role = system, role
        _chk(lib().lgc_party_create(C.byref(self._h), device, C.byref(system), role, seed, max_launch_table_bytes))
        self.num_launches = lib().lgc_party_num_launches(self._h)
        self.input_bits = lib().lgc_party_input_bits(self._h)
        self.num_reveal = lib().lgc_party_num_reveal(self._h)
        self.and_gates = lib().lgc_party_and_gates(self._h)

    def table_bytes(self, k):
        return lib().lgc_party_table_bytes(self._h, k)

    def program_fingerprint(self):
        """32 bytes over everything the two roles must agree on (lgc_party_program_fingerprint)"""
        out = np.zeros(32, dtype=np.uint8)
        L = lib()
        L.lgc_party_program_fingerprint.argtypes = [C.c_void_p, C.c_void_p]; L.lgc_party_program_fingerprint.restype = C.c_int
        _chk(L.lgc_party_program_fingerprint(self._h, _vp(out)))
        return out.tobytes()

    def input_pairs(self, share):
        m0 = np.zeros((self.input_bits, 16), dtype=np.uint8); m1 = np.zeros((self.input_bits, 16), dtype=np.uint8)
        _chk(lib().lgc_party_input_pairs(self._h, share, _vp(m0), _vp(m1)))
        return m0, m1

    def encode_inputs(self, share, values):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros((self.input_bits, 16), dtype=np.uint8)
        _chk(lib().lgc_party_encode_inputs(self._h, share, _vp(values), _vp(out)))
        return out

    def set_input_labels(self, share, labels):
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        assert labels.size == self.input_bits * 16
        _chk(lib().lgc_party_set_input_labels(self._h, share, _vp(labels)))

    def garble(self, k):
        buf = np.zeros(max(1, self.table_bytes(k)), dtype=np.uint8)
        _chk(lib().lgc_party_garble(self._h, k, _vp(buf)))
        return buf[:self.table_bytes(k)]

    def evaluate(self, k, tables):
        tables = np.ascontiguousarray(tables, dtype=np.uint8)
        _chk(lib().lgc_party_evaluate(self._h, k, _vp(tables) if tables.size else None))

    # ---- device-resident table ring (garbler and evaluator processes on one node)
    def ring_create(self, nslots):
        """garbler: allocate the ring; returns (64-byte hipIpc handle, slot bytes) for the evaluator process"""
        h = np.zeros(64, dtype=np.uint8); sb = C.c_size_t()
        _chk(lib().lgc_party_ring_create(self._h, nslots, _vp(h), C.byref(sb)))
        return h.tobytes(), sb.value

    def ring_open(self, handle, nslots, slot_bytes):
        _chk(lib().lgc_party_ring_open(self._h, _vp(np.frombuffer(handle, dtype=np.uint8).copy()), nslots, slot_bytes))

    def garble_ring(self, k):
        _chk(lib().lgc_party_garble_ring(self._h, k))

    def evaluate_ring(self, k):
        _chk(lib().lgc_party_evaluate_ring(self._h, k))

    def garble_ring_begin(self, k):
        """garbler: enqueue launch k into the ring and return at once (lgc_party_garble_ring_begin)"""
        _chk(lib().lgc_party_garble_ring_begin(self._h, k))

    def garble_ring_streams(self, n):
        """1: table passes on the record kernels' stream (before the first garble_ring_begin); 2: a stream of their own"""
        _chk(lib().lgc_party_garble_ring_streams(self._h, n))

    def garble_ring_wait(self, k):
        """garbler: return once the tables of launch k are complete in the ring"""
        _chk(lib().lgc_party_garble_ring_wait(self._h, k))

    def test_garble_ring_stage(self, k, stage):
        """test hook: stage 1 = record kernel of launch k into the ring path, 2 = its table pass; returns True when the
        launch is garbled on the critical path (has a table pass)"""
        crit = C.c_int()
        _chk(lib().lgc_test_party_garble_ring_stage(self._h, k, stage, C.byref(crit)))
        return bool(crit.value)

    def test_ring_read(self, k, nbytes):
        out = np.zeros(max(1, nbytes), dtype=np.uint8)
        _chk(lib().lgc_test_party_ring_read(self._h, k, _vp(out), nbytes))
        return out[:nbytes]

    def decode_bits(self):
        out = np.zeros(max(1, self.num_reveal), dtype=np.uint64)
        _chk(lib().lgc_party_decode_bits(self._h, _vp(out)))
        return out

    def finish(self, garbler_dec):
        d = self.system.d
        beta = np.zeros(d, dtype=np.int64)
        trace = np.zeros((max(1, self.system.num_iterations), d + 4), dtype=np.int64)
        inputs = np.zeros(d * (d + 1) // 2 + d, dtype=np.int64)
        _chk(lib().lgc_party_finish(self._h, _vp(np.ascontiguousarray(garbler_dec, dtype=np.uint64)), _vp(beta), _vp(trace), _vp(inputs)))
        return beta, trace, inputs

    def close(self):
        if self._h:
            lib().lgc_party_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def share_choice_bits(values, width):
    """sel[i*intsize+j] = (input[i] >> j) & 1  (reference src/input.c:41)"""
    values = np.ascontiguousarray(values, dtype=np.uint64)
    return ((values[:, None] >> np.arange(width, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).ravel()


class OtSender:
    """IKNP extension sender (holds delta and the 128 seeds k_j^{delta_j} from the base OTs)"""

    def __init__(self, delta, seeds, device=0):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint8).reshape(128, 16)
        self._h = C.c_void_p()
        _chk(lib().lgc_ot_sender_create(C.byref(self._h), device, bytes(delta), _vp(seeds)))

    def gilboa(self, b, width, u):
        b = np.ascontiguousarray(b, dtype=np.uint64); npairs, n = b.shape
        y = np.zeros(npairs * n * width, dtype=np.uint64); sh = np.zeros(npairs, dtype=np.uint64)
        _chk(lib().lgc_ot_gilboa_send(self._h, _vp(b), npairs, n, width, _vp(u), _vp(y), _vp(sh)))
        return y, sh

    # raw-pointer forms (integer addresses): page-locked host buffers (host_alloc) or, after
    # set_device_io(True), device memory used in place
    def set_device_io(self, on):
        _chk(lib().lgc_ot_sender_set_device_io(self._h, 1 if on else 0))

    def gilboa_ptr(self, b_ptr, npairs, n, width, u_ptr, y_ptr, shares_ptr):
        _chk(lib().lgc_ot_gilboa_send(self._h, C.c_void_p(b_ptr), npairs, n, width, C.c_void_p(u_ptr), C.c_void_p(y_ptr), C.c_void_p(shares_ptr)))

    def labels(self, m0, m1, u):
        m0 = np.ascontiguousarray(m0, dtype=np.uint8).reshape(-1, 16); m1 = np.ascontiguousarray(m1, dtype=np.uint8).reshape(-1, 16)
        e = np.zeros((len(m0), 32), dtype=np.uint8)
        _chk(lib().lgc_ot_labels_send(self._h, _vp(m0), _vp(m1), len(m0), _vp(u), _vp(e)))
        return e

    def close(self):
        if self._h:
            lib().lgc_ot_sender_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OtReceiver:
    """IKNP extension receiver (holds the 128 seed pairs (k_j^0, k_j^1) from the base OTs)"""

    def __init__(self, seeds0, seeds1, device=0):
        seeds0 = np.ascontiguousarray(seeds0, dtype=np.uint8).reshape(128, 16)
        seeds1 = np.ascontiguousarray(seeds1, dtype=np.uint8).reshape(128, 16)
        self._h = C.c_void_p()
        _chk(lib().lgc_ot_receiver_create(C.byref(self._h), device, _vp(seeds0), _vp(seeds1)))

    def gilboa_start(self, a, width):
        a = np.ascontiguousarray(a, dtype=np.uint64); npairs, n = a.shape
        u = np.zeros(lib().lgc_ot_u_bytes(npairs * n * width), dtype=np.uint8)
        _chk(lib().lgc_ot_gilboa_recv_start(self._h, _vp(a), npairs, n, width, _vp(u)))
        if not hasattr(self, "_nps"):
            self._nps = []
        self._nps.append(npairs)               # several receives may be in flight; finishes complete the oldest
        return u

    def set_device_io(self, on):
        _chk(lib().lgc_ot_receiver_set_device_io(self._h, 1 if on else 0))

    def gilboa_start_ptr(self, a_ptr, npairs, n, width, u_ptr):
        _chk(lib().lgc_ot_gilboa_recv_start(self._h, C.c_void_p(a_ptr), npairs, n, width, C.c_void_p(u_ptr)))

    def gilboa_finish_ptr(self, y_ptr, shares_ptr):
        _chk(lib().lgc_ot_gilboa_recv_finish(self._h, C.c_void_p(y_ptr), C.c_void_p(shares_ptr)))

    def gilboa_finish(self, y):
        if not getattr(self, "_nps", None):
            sh = np.zeros(1, dtype=np.uint64)
            _chk(lib().lgc_ot_gilboa_recv_finish(self._h, _vp(np.ascontiguousarray(y, dtype=np.uint64)), _vp(sh)))
        sh = np.zeros(self._nps.pop(0), dtype=np.uint64)
        _chk(lib().lgc_ot_gilboa_recv_finish(self._h, _vp(np.ascontiguousarray(y, dtype=np.uint64)), _vp(sh)))
        return sh

    def labels_start(self, choice):
        choice = np.ascontiguousarray(choice, dtype=np.uint8)
        self._m = len(choice)
        u = np.zeros(lib().lgc_ot_u_bytes(self._m), dtype=np.uint8)
        _chk(lib().lgc_ot_labels_recv_start(self._h, _vp(choice), self._m, _vp(u)))
        return u

    def labels_finish(self, e):
        out = np.zeros((self._m, 16), dtype=np.uint8)
        _chk(lib().lgc_ot_labels_recv_finish(self._h, _vp(np.ascontiguousarray(e, dtype=np.uint8)), _vp(out)))
        return out

    def close(self):
        if self._h:
            lib().lgc_ot_receiver_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_cached_memory():
    """free the table ring a closed Solver left parked for the next one"""
    lib().lgc_release_cached_memory()


def set_split_kernels(garbler=True, evaluator=True):
    """which kernel runs the latency-bound launches of each role (16-wave column-split or 4-wave); interchangeable"""
    lib().lgc_set_split_kernels(int(bool(garbler)), int(bool(evaluator)))


def reference_gate_count(algorithm, width, d, iterations=0):
    """AND gates of the reference's circuit for the same solve, or None where it published none"""
    L = lib()
    L.lgc_reference_gate_count.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    L.lgc_reference_gate_count.restype = C.c_int
    g = C.c_uint64()
    alg = ALG[algorithm] if isinstance(algorithm, str) else int(algorithm)
    return int(g.value) if L.lgc_reference_gate_count(alg, width, d, iterations, C.byref(g)) == 0 else None


def set_karatsuba(on=True):
    """Karatsuba products in the CGD matrix-vector launches (width 64) for programs built from now on"""
    lib().lgc_set_karatsuba.argtypes = [C.c_int]; lib().lgc_set_karatsuba.restype = None
    lib().lgc_set_karatsuba(int(bool(on)))


def set_table_ring_slack(nbytes):
    """room in a co-located solver's table ring beyond its largest launch (0: the default, 8 GiB)"""
    lib().lgc_set_table_ring_slack.argtypes = [C.c_size_t]; lib().lgc_set_table_ring_slack.restype = None
    lib().lgc_set_table_ring_slack(int(nbytes))




def devices_preflight(devices):
    """lgc_devices_preflight (linreg_gc_sweep.h): every index exists and distinct devices can reach each other; raises LgcError
    with the index or the pair in the message otherwise"""
    arr = (C.c_int * len(devices))(*[int(v) for v in devices])
    L = lib()
    L.lgc_devices_preflight.argtypes = [C.c_void_p, C.c_size_t]; L.lgc_devices_preflight.restype = C.c_int
    _chk(L.lgc_devices_preflight(arr, len(devices)))


def gate_hash_eval(labels, tweaks, device=0):
    """the gate hash H(x, t) on the device: labels (n, 16) uint8, tweaks (n,) uint64 -> (n, 16) uint8"""
    x = np.ascontiguousarray(labels, dtype=np.uint8).reshape(-1, 16)
    t = np.ascontiguousarray(tweaks, dtype=np.uint64).reshape(-1)
    assert len(t) == len(x)
    out = np.zeros_like(x)
    L = lib()
    L.lgc_gate_hash_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]; L.lgc_gate_hash_eval.restype = C.c_int
    _chk(L.lgc_gate_hash_eval(device, x.ctypes.data, t.ctypes.data, out.ctypes.data, len(x)))
    return out


def aes_bench(waves=8192, blocks_per_lane=256, device=0):
    rate, chk = C.c_double(), C.c_uint32()
    _chk(lib().lgc_aes_bench(device, waves, blocks_per_lane, C.byref(rate), C.byref(chk)))
    return rate.value, chk.value


def aes_encrypt(blocks, device=0):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.empty_like(blocks)
    _chk(lib().lgc_aes_encrypt(device, blocks.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(blocks)))
    return out
