#!/usr/bin/env python3
"""Experiment (not product code): a bitsliced AES S-box mapped onto 3-input LUTs (v_bitop3_b32 on gfx950).

Source circuit: Boyar & Peralta, "A small depth-16 circuit for the AES S-box" (2012) -- 128 two-input gates (XOR / XNOR /
AND), restated here from the paper and checked exhaustively against the S-box computed from first principles (inverse in
GF(2^8) + affine map).  The mapper enumerates 3-feasible cuts of the gate DAG and covers it by area flow + a few rounds
of exact-area refinement; every cell's 8-bit truth table is derived by simulation and the mapped netlist is checked on all
256 inputs again.  Output: a C++ include with one function over eight 32-bit bit planes (32 blocks per plane).

    python3 scripts/exp/gen_bsaes.py > scripts/exp/bsaes_sbox.inc
"""
import itertools
import random
import sys

BP = """
T1 = U0 + U3
T2 = U0 + U5
T3 = U0 + U6
T4 = U3 + U5
T5 = U4 + U6
T6 = T1 + T5
T7 = U1 + U2
T8 = U7 + T6
T9 = U7 + T7
T10 = T6 + T7
T11 = U1 + U5
T12 = U2 + U5
T13 = T3 + T4
T14 = T6 + T11
T15 = T5 + T11
T16 = T5 + T12
T17 = T9 + T16
T18 = U3 + U7
T19 = T7 + T18
T20 = T1 + T19
T21 = U6 + U7
T22 = T7 + T21
T23 = T2 + T22
T24 = T2 + T10
T25 = T20 + T17
T26 = T3 + T16
T27 = T1 + T12
M1 = T13 x T6
M2 = T23 x T8
M3 = T14 + M1
M4 = T19 x U7
M5 = M4 + M1
M6 = T3 x T16
M7 = T22 x T9
M8 = T26 + M6
M9 = T20 x T17
M10 = M9 + M6
M11 = T1 x T15
M12 = T4 x T27
M13 = M12 + M11
M14 = T2 x T10
M15 = M14 + M11
M16 = M3 + M2
M17 = M5 + T24
M18 = M8 + M7
M19 = M10 + M15
M20 = M16 + M13
M21 = M17 + M15
M22 = M18 + M13
M23 = M19 + T25
M24 = M22 + M23
M25 = M22 x M20
M26 = M21 + M25
M27 = M20 + M21
M28 = M23 + M25
M29 = M28 x M27
M30 = M26 x M24
M31 = M20 x M23
M32 = M27 x M31
M33 = M27 + M25
M34 = M21 x M22
M35 = M24 x M34
M36 = M24 + M25
M37 = M21 + M29
M38 = M32 + M33
M39 = M23 + M30
M40 = M35 + M36
M41 = M38 + M40
M42 = M37 + M39
M43 = M37 + M38
M44 = M39 + M40
M45 = M42 + M41
M46 = M44 x T6
M47 = M40 x T8
M48 = M39 x U7
M49 = M43 x T16
M50 = M38 x T9
M51 = M37 x T17
M52 = M42 x T15
M53 = M45 x T27
M54 = M41 x T10
M55 = M44 x T13
M56 = M40 x T23
M57 = M39 x T19
M58 = M43 x T3
M59 = M38 x T22
M60 = M37 x T20
M61 = M42 x T1
M62 = M45 x T4
M63 = M41 x T2
L0 = M61 + M62
L1 = M50 + M56
L2 = M46 + M48
L3 = M47 + M55
L4 = M54 + M58
L5 = M49 + M61
L6 = M62 + L5
L7 = M46 + L3
L8 = M51 + M59
L9 = M52 + M53
L10 = M53 + L4
L11 = M60 + L2
L12 = M48 + M51
L13 = M50 + L0
L14 = M52 + M61
L15 = M55 + L1
L16 = M56 + L0
L17 = M57 + L1
L18 = M58 + L8
L19 = M63 + L4
L20 = L0 + L1
L21 = L1 + L7
L22 = L3 + L12
L23 = L18 + L2
L24 = L15 + L9
L25 = L6 + L10
L26 = L7 + L9
L27 = L8 + L10
L28 = L11 + L14
L29 = L11 + L17
S0 = L6 + L24
S1 = L16 # L26
S2 = L19 # L28
S3 = L6 + L21
S4 = L20 + L22
S5 = L25 + L29
S6 = L13 # L27
S7 = L6 # L23
"""


def sbox_table():
    def xtime(v):
        return ((v << 1) ^ (0x1b if v & 0x80 else 0)) & 0xff
    pw, lg = [0] * 256, [0] * 256
    g = 1
    for i in range(255):
        pw[i] = g
        lg[g] = i
        g ^= xtime(g)
    out = []
    for x in range(256):
        inv = pw[(255 - lg[x]) % 255] if x else 0
        s = r = inv
        for _ in range(4):
            r = ((r << 1) | (r >> 7)) & 0xff
            s ^= r
        out.append(s ^ 0x63)
    return out


def parse():
    gates = []   # (name, op, x, y)
    for line in BP.strip().splitlines():
        name, rhs = [t.strip() for t in line.split("=")]
        x, op, y = rhs.split()
        gates.append((name, op, x, y))
    return gates


INPUTS = ["U%d" % i for i in range(8)]
OUTPUTS = ["S%d" % i for i in range(8)]
# all 256 input bytes at once: value of a signal = 256-bit integer (bit v = the signal's value on input byte v)
ALL = (1 << 256) - 1


def input_vals():
    vals = {}
    for i in range(8):   # U0 = most significant bit of the byte
        m = 0
        for v in range(256):
            if (v >> (7 - i)) & 1:
                m |= 1 << v
        vals["U%d" % i] = m
    return vals


def simulate(gates):
    vals = input_vals()
    for name, op, x, y in gates:
        a, b = vals[x], vals[y]
        vals[name] = (a ^ b) if op == "+" else ((a ^ b) ^ ALL) if op == "#" else (a & b)
    return vals


def check_against_sbox(vals):
    sb = sbox_table()
    for v in range(256):
        got = 0
        for i in range(8):
            got |= ((vals["S%d" % i] >> v) & 1) << (7 - i)
        assert got == sb[v], (v, got, sb[v])


def map_lut3(gates, seed):
    rnd = random.Random(seed)
    node = {g[0]: g for g in gates}
    order = [g[0] for g in gates]
    fanout = {n: 0 for n in INPUTS + order}
    for _, _, x, y in gates:
        fanout[x] += 1
        fanout[y] += 1
    for o in OUTPUTS:
        fanout[o] += 1
    # 3-feasible cuts
    cuts = {n: [frozenset([n])] for n in INPUTS}
    for name, _, x, y in gates:
        cs = {frozenset([name])}
        for cx in cuts[x]:
            for cy in cuts[y]:
                u = cx | cy
                if len(u) <= 3:
                    cs.add(u)
        cuts[name] = list(cs)
    # area flow
    af = {n: 0.0 for n in INPUTS}
    best = {}
    for name in order:
        cand = []
        for c in cuts[name]:
            if c == frozenset([name]):
                continue
            cost = 1.0 + sum(af[l] / max(1, fanout[l]) for l in c) + rnd.random() * 0.05
            cand.append((cost, sorted(c)))
        cand.sort()
        af[name] = cand[0][0]
        best[name] = frozenset(cand[0][1])

    def cover(best):
        need, stack = set(), list(OUTPUTS)
        while stack:
            n = stack.pop()
            if n in need or n in INPUTS:
                continue
            need.add(n)
            stack.extend(best[n])
        return need

    need = cover(best)
    # exact-area refinement: for a mapped node try every other cut, keep the one with the smallest cover
    for _ in range(6):
        improved = False
        names = [n for n in order if n in need]
        rnd.shuffle(names)
        for n in names:
            cur = len(need)
            keep = best[n]
            for c in cuts[n]:
                if c == frozenset([n]) or c == keep:
                    continue
                best[n] = c
                m = cover(best)
                if len(m) < cur:
                    cur, keep, improved = len(m), c, True
            best[n] = keep
            need = cover(best)
        if not improved:
            break
    return need, best


def truth_table(root, leaves, node):
    """8-bit table of `root` as a function of (a, b, c) = leaves, index = a<<2 | b<<1 | c (v_bitop3 convention)."""
    leaves = list(leaves) + [None] * (3 - len(leaves))
    tt = 0
    for idx in range(8):
        env = {}
        for k, l in enumerate(leaves):
            if l is not None:
                env[l] = (idx >> (2 - k)) & 1

        def ev(n):
            if n in env:
                return env[n]
            _, op, x, y = node[n]
            a, b = ev(x), ev(y)
            r = (a ^ b) if op == "+" else (a ^ b ^ 1) if op == "#" else (a & b)
            env[n] = r
            return r
        tt |= ev(root) << idx
    return leaves, tt


def emit(gates, need, best):
    node = {g[0]: g for g in gates}
    order = [g[0] for g in gates if g[0] in need]
    cells = []
    for n in order:
        leaves, tt = truth_table(n, sorted(best[n]), node)
        cells.append((n, leaves, tt))
    # check the mapped netlist on all 256 inputs
    vals = input_vals()
    for n, leaves, tt in cells:
        r = 0
        a, b, c = [(vals[l] if l is not None else 0) for l in leaves]
        for idx in range(8):
            if (tt >> idx) & 1:
                ta = a if idx & 4 else a ^ ALL
                tb = b if idx & 2 else b ^ ALL
                tc = c if idx & 1 else c ^ ALL
                r |= ta & tb & tc
        vals[n] = r
    check_against_sbox(vals)
    out = []
    out.append("// generated by scripts/exp/gen_bsaes.py -- do not edit.  %d three-input cells (Boyar-Peralta depth-16 circuit, 128 gates)" % len(cells))
    out.append("// u[0] = most significant bit plane of the byte ... u[7] = least significant; in place.")
    out.append("#define BS_SBOX_CELLS %d" % len(cells))
    out.append("static BS_HD void bs_sbox(uint32_t *u) {")
    out.append("    const uint32_t U0 = u[0], U1 = u[1], U2 = u[2], U3 = u[3], U4 = u[4], U5 = u[5], U6 = u[6], U7 = u[7];")
    for n, leaves, tt in cells:
        args = [l if l is not None else "0u" for l in leaves]
        nl = sum(1 for l in leaves if l is not None)
        if nl == 2 and (tt & 0xff) in (0x3c,):   # plain xor of a, b (c unused: table independent of c)
            out.append("    const uint32_t %s = %s ^ %s;" % (n, args[0], args[1]))
        elif nl == 2 and (tt & 0xff) == 0xc0:
            out.append("    const uint32_t %s = %s & %s;" % (n, args[0], args[1]))
        else:
            out.append("    const uint32_t %s = BS_LUT(%s, %s, %s, 0x%02x);" % (n, args[0], args[1], args[2], tt))
    out.append("    " + " ".join("u[%d] = S%d;" % (i, i) for i in range(8)))
    out.append("}")
    return "\n".join(out), len(cells)


def main():
    gates = parse()
    assert len(gates) == 128
    check_against_sbox(simulate(gates))
    bestn, bestout = None, None
    for seed in range(40):
        need, best = map_lut3(gates, seed)
        text, n = emit(gates, need, best)
        if bestn is None or n < bestn:
            bestn, bestout = n, text
    sys.stderr.write("S-box: %d LUT3 cells\n" % bestn)
    print(bestout)


if __name__ == "__main__":
    main()
