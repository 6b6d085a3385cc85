#!/usr/bin/env python3
"""Writes multi_orbslam3_amd/csrc/ldlt_jump_tables.inc: the assembly text of k_ldlt_big's per-tile dispatch.

A tile of the factorisation lives in fixed registers (tile S < 32: a[8S:8S+7], S >= 32: v[128+8(S-32) : +7]); the code that
touches tile S therefore differs from tile to tile only in register numbers.  hipcc lowers a switch over S to a chain of
up to 48 compare-and-branch blocks (500-900 cycles per dispatch, measured); here the dispatch is a computed jump
(s_setpc_b64) into a table of equally sized cases, all inside one assembly statement.

Operands (outputs are numbered first):
           MFMA4: %0 = scratch scalar (output), %1..%8 = a0,w0,a1,w1,a2,w2,a3,w3 (64-bit vector registers), %9 = slot (scalar)
           MFMA8: %0 = scratch scalar (output), %1..%16 operands of tile S (8) then S+1 (8), %17 = pair index (scalar)
           GET:   %0..%7 = outputs (32-bit vector registers), %8 = scratch scalar (output), %9 = slot (scalar)
"""
import os

def treg(s, lo, hi=None):
    f, b = ("a", 8 * s) if s < 32 else ("v", 128 + 8 * (s - 32))
    return f"{f}[{b + lo}:{b + hi}]" if hi is not None else f"{f}{b + lo}"

def prologue(idx, tmp, stride):
    # s_getpc_b64 returns the address of the instruction after it; the five instructions up to the first case are
    # 4 bytes each, except s_mul_i32 with a literal > 64 (8 bytes)
    mul_bytes = 4 if stride <= 64 else 8
    k = mul_bytes + 4 + 4 + 4 + 4
    return [f"s_getpc_b64 vcc", f"s_mul_i32 {tmp}, {idx}, {stride}", f"s_add_u32 {tmp}, {tmp}, {k}",
            f"s_add_u32 vcc_lo, vcc_lo, {tmp}", "s_addc_u32 vcc_hi, vcc_hi, 0", "s_setpc_b64 vcc"]

def table(cases, stride, case_bytes):
    out = []
    for body, nbytes in zip(cases, case_bytes):
        pad = stride - nbytes - 4
        assert pad >= 0 and pad % 4 == 0
        out += body + ["s_branch .Lend_%="] + ["s_nop 0"] * (pad // 4)
    out.append(".Lend_%=:")
    return out

def mfma4(n):
    cases = [[f"v_mfma_f64_16x16x4_f64 {treg(s,0,7)}, %{1+2*q}, %{2+2*q}, {treg(s,0,7)}" for q in range(4)] for s in range(n)]
    return prologue("%9", "%0", 40) + table(cases, 40, [32] * n)

def mfma8(n):
    cases = []
    for g in range(n // 2):
        c = []
        for q in range(4):
            c.append(f"v_mfma_f64_16x16x4_f64 {treg(2*g,0,7)}, %{1+2*q}, %{2+2*q}, {treg(2*g,0,7)}")
            c.append(f"v_mfma_f64_16x16x4_f64 {treg(2*g+1,0,7)}, %{9+2*q}, %{10+2*q}, {treg(2*g+1,0,7)}")
        cases.append(c)
    return prologue("%17", "%0", 72) + table(cases, 72, [64] * (n // 2))

# ---- BULK: the trailing update's steady state as ONE block of straight-line code per pair of tiles, entered by a computed jump at
# an EVEN pair and left after %8 pairs.  A lone wavefront issues an instruction every ~5 cycles and a matrix instruction keeps
# its pipe for 64: everything that is not a matrix instruction -- the next pair's tile coordinates out of the slot table
# (v_readlane with a constant lane), its LDS addresses, its eight 128-bit operand reads, the loop control -- sits BETWEEN the
# eight matrix instructions of the current pair (in front of them it cost ~200 cycles per pair: 354 cycles per tile for 256).
# Operand sets in fixed registers: A = v[64:95], B = v[96:127] (per tile of the pair 16 registers: a01 a23 w01 w23), addresses
# v[60:63]; even pairs compute on A and fetch into B, odd pairs the other way round.
#   %0 %1 %2 %3 = scalar scratch (early-clobber outputs), %4 = slot table (vector), %5 = LDS byte address of this lane's part of
#   the -R images of the row's parity, %6 = the same of the W images, %7 = first pair (scalar, even), %8 = pairs to do (scalar, > 0)
SETS = (64, 96)
def opnd(setbase, t, q):      # A / W operand q of tile t of the pair
    b = setbase + 16 * t
    return f"v[{b + 2*q}:{b + 2*q + 1}]", f"v[{b + 8 + 2*q}:{b + 8 + 2*q + 1}]"
def fetch(setbase):           # eight reads: (dst, address register, offset)
    out = []
    for t in range(2):
        b = setbase + 16 * t
        out += [(f"v[{b}:{b+3}]", f"v{60 + 2*t}", 0), (f"v[{b+4}:{b+7}]", f"v{60 + 2*t}", 1024),
                (f"v[{b+8}:{b+11}]", f"v{61 + 2*t}", 0), (f"v[{b+12}:{b+15}]", f"v{61 + 2*t}", 1024)]
    return [f"ds_read_b128 {d}, {a}" + (f" offset:{o}" if o else "") for d, a, o in out]
def addr_math(sreg, t):       # tile coordinates (i | j << 8) in sreg -> v[60+2t] (-R image of column i), v[61+2t] (W image of column j)
    return [f"s_and_b32 %2, {sreg}, 0xff", f"s_lshr_b32 {sreg}, {sreg}, 8",
            f"v_lshl_add_u32 v{60 + 2*t}, %2, 12, %5", f"v_lshl_add_u32 v{61 + 2*t}, {sreg}, 12, %6"]
def bulk(n):
    npairs = n // 2
    L = ["s_mov_b32 %3, %8", "s_lshl_b32 %2, %7, 1", "s_nop 3", "v_readlane_b32 %0, %4, %2", "s_add_u32 %2, %2, 1", "s_nop 3",
         "v_readlane_b32 %1, %4, %2"]
    L += addr_math("%0", 0) + addr_math("%1", 1) + fetch(SETS[0])
    L += ["s_getpc_b64 vcc", ".Lpc_%=:", "s_lshl_b32 %2, %7, 8", "s_add_u32 %2, %2, .Lblk0_%=-.Lpc_%=", "s_add_u32 vcc_lo, vcc_lo, %2",
          "s_addc_u32 vcc_hi, vcc_hi, 0", "s_setpc_b64 vcc"]
    for p in range(npairs):
        cur, nxt = SETS[p & 1], SETS[(p + 1) & 1]
        m = []
        for q in range(4):
            for t in range(2):
                a, w = opnd(cur, t, q)
                m.append(f"v_mfma_f64_16x16x4_f64 {treg(2*p+t,0,7)}, {a}, {w}, {treg(2*p+t,0,7)}")
        f = fetch(nxt)
        B = [".p2align 8", f".Lblk{p}_%=:", "s_waitcnt lgkmcnt(0)",
             f"v_readlane_b32 %0, %4, {2*p+2}", f"v_readlane_b32 %1, %4, {2*p+3}", m[0]]
        # (the reads early in the block: issued behind the fifth matrix instruction the last of them were still in flight when the
        # next block asked for them -- ~100 cycles per pair with four wavefronts on the LDS)
        B += addr_math("%0", 0) + addr_math("%1", 1) + [m[1]] + f[0:3] + [m[2]] + f[3:6] + [m[3]] + f[6:8] + [m[4], m[5]]
        B += [m[6], "s_sub_u32 %3, %3, 1", "s_cmp_eq_u32 %3, 0", m[7], "s_cbranch_scc1 .Lend_%="]
        if p + 1 < npairs:
            B.append(f"s_branch .Lblk{p+1}_%=")
        L += B
    L += [".Lend_%=:", "s_waitcnt lgkmcnt(0)"]
    return L

def get(n, wait=True):
    cases, nb = [], []
    for s in range(n):
        if s < 32:
            cases.append([f"v_accvgpr_read_b32 %{i}, {treg(s,i)}" for i in range(8)]); nb.append(64)
        else:
            cases.append([f"v_mov_b32 %{i}, {treg(s,i)}" for i in range(8)]); nb.append(32)
    # 18 wait states between a matrix instruction's write and a read of its result (wait = False: the caller knows that the last
    # matrix instruction on the tile is further away)
    return (["s_nop 15", "s_nop 7"] if wait else []) + prologue("%9", "%8", 72) + table(cases, 72, nb)

def cstr(lines):
    return " \\\n".join('  "' + l + '\\n\\t"' for l in lines)

def render():
    out = ["// GENERATED by tools/gen/gen_ldlt_jump_tables.py -- assembly text of k_ldlt_big's per-tile dispatch (see there)\n"]
    for n in (32, 48):
        out.append(f"#define LDLTM_JT_MFMA4_{n} \\\n{cstr(mfma4(n))}\n")
        out.append(f"#define LDLTM_JT_MFMA8_{n} \\\n{cstr(mfma8(n))}\n")
        out.append(f"#define LDLTM_JT_GET_{n} \\\n{cstr(get(n))}\n")
        out.append(f"#define LDLTM_JT_GETNW_{n} \\\n{cstr(get(n, False))}\n")
        out.append(f"#define LDLTM_JT_BULK_{n} \\\n{cstr(bulk(n))}\n")
    return "".join(out)

DST = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multi_orbslam3_amd", "csrc",
                                    "ldlt_jump_tables.inc"))

if __name__ == "__main__":
    with open(DST, "w") as f:
        f.write(render())
    print("wrote", DST)
