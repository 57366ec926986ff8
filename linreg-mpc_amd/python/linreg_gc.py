"""ctypes binding of the product C ABI (include/linreg_gc.h, liblinreg_gc.so).

This is plumbing above the C ABI; all compute happens in the HIP library.
There is no CPU fallback: `Solver` raises `LgcError` (LGC_ENODEVICE) when no
MI355X is visible, and importing fails loudly when the library is not built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "liblinreg_gc.so")

ALG = {"cholesky": 0, "ldlt": 1, "cgd": 2}


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
                ("total_gates", C.c_uint64), ("max_launch_steps", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C linreg-mpc_amd/csrc`)" % LIB_PATH)
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
            ("lgc_solver_get_profile", [vp, vp, vp, sz]),
            ("lgc_program_build", [C.POINTER(vp), C.POINTER(System)]),
            ("lgc_program_info_get", [vp, C.POINTER(ProgramInfo)]),
            ("lgc_aes_bench", [ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
            ("lgc_aes_encrypt", [ci, vp, vp, sz]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = ci, args
        L.lgc_solver_destroy.argtypes = [vp]; L.lgc_solver_destroy.restype = None
        L.lgc_program_destroy.argtypes = [vp]; L.lgc_program_destroy.restype = None
        L.lgc_program_records.argtypes = [vp]; L.lgc_program_records.restype = C.POINTER(Record)
        L.lgc_program_launches.argtypes = [vp]; L.lgc_program_launches.restype = C.POINTER(Launch)
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise LgcError(rc, lib().lgc_last_error().decode())


def device_count():
    return lib().lgc_device_count()


def make_system(d, width=64, precision=56, algorithm="cgd", num_iterations=0, lam=0.0, nshares=2,
                normalize=0, reveal_inputs=0, trace=0):
    alg = ALG[algorithm] if isinstance(algorithm, str) else int(algorithm)
    return System(d, width, precision, alg, num_iterations, lam, nshares, normalize, reveal_inputs, trace)


class Program:
    """The lowered circuit program (host only; needs no GPU)."""

    def __init__(self, system):
        self._h = C.c_void_p()
        _chk(lib().lgc_program_build(C.byref(self._h), C.byref(system)))
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

    def close(self):
        if self._h:
            lib().lgc_program_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Solver:
    """Garble + evaluate one linear system on one MI355X (both roles co-located).

    Replaces `execYaoProtocol(pd, solver, &ls)` (reference src/cmd/linreg.c:177)."""

    def __init__(self, system, seed=b"\x01" * 16, device=0):
        assert len(seed) == 16
        self._h = C.c_void_p()
        self.system = system
        _chk(lib().lgc_solver_create(C.byref(self._h), device, C.byref(system), seed))

    def set_shares(self, shares):
        d = self.system.d
        shares = np.ascontiguousarray(shares, dtype=np.uint64)
        assert shares.size == self.system.nshares * (d * (d + 1) // 2 + d), shares.shape
        _chk(lib().lgc_solver_set_shares(self._h, shares.ctypes.data_as(C.c_void_p)))

    def run(self, profile=False):
        _chk(lib().lgc_solver_run(self._h, 1 if profile else 0))

    def beta(self):
        out = np.zeros(self.system.d, dtype=np.int64)
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

    def close(self):
        if self._h:
            lib().lgc_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aes_bench(waves=8192, blocks_per_lane=256, device=0):
    rate, chk = C.c_double(), C.c_uint32()
    _chk(lib().lgc_aes_bench(device, waves, blocks_per_lane, C.byref(rate), C.byref(chk)))
    return rate.value, chk.value


def aes_encrypt(blocks, device=0):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.empty_like(blocks)
    _chk(lib().lgc_aes_encrypt(device, blocks.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(blocks)))
    return out
