#!/usr/bin/env python3
"""Generator of the hand-scheduled tile loop of the attention forward (csrc/attn_fwd_asm.inc; k_attn_fwd_asm in attn_bf16.hip).
Reference operator: F.scaled_dot_product_attention(q, k, v, dropout_p), src/model/layers/attn.py:122-127.

One workgroup = 4 waves = ONE wave per SIMD; a wave owns QT = 4 query tiles (128 queries) and walks the stage's KT = 4 key tiles.
Per (key tile, query tile) UNIT the arithmetic is exactly the bound-based tile of k_attn_fwd_bf16<4, 4, DROP, true, 8>: S^T = K Q^T
(2 MFMAs, C = 0), p = exp2(S) (16 v_exp_f32: the workgroup's rows satisfy |q|^2 max|k|^2 <= 54^2, no reference value), 8 packs, the
row sums of the PACKED p (8 v_dot2c_f32_bf16), the dropout mask on the packed pairs (xor, saturating subtract, shift, and), O^T += V^T
P^T (2 MFMAs).  What one wave per SIMD buys over the compiled four-waves-per-SIMD kernel:
  * the K / V^T fragments and the mask's column words of a key tile are read from LDS ONCE for the four query tiles (2 LDS
    instructions per unit instead of 8; an LDS instruction costs ~10 cycles of the port the vector stream needs);
  * the Q fragments (a0-a31) and the four O^T accumulators (a32-a95) live in AGPRs for the whole launch, the score tile is double
    buffered, and every MFMA sits in the vector stream with >= 8 instructions (32 issue cycles) behind it;
  * LDS reads are issued a key tile ahead and waited for by count (the Stream class of gen_attn_bwd_asm.py).
Budget per unit (tools/lab/inst_cost.hip): 16 x 8 + 8 x 5 + 8 x 4 + 8 x (4 + 5 + 5 + 4) = 344 vector cycles + 4 x 8 MFMA issue + ~20 LDS.

Usage: python3 gen_attn_fwd_asm.py > attn_fwd_asm.inc   (tests/test_host_cpu.py checks that the committed file is current).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_bwd_asm import Stream, vr, ar, c_string   # noqa: E402

QT = 4            # query tiles per wave
KT = int(os.environ.get("GEN_FWD_KT", "8"))            # key tiles per stage (256 keys: the stage code -- two barriers, LDS
                  # staging, the loop's prologue -- is ~700 cycles per stage)
TILE = 2048
MFMA = "v_mfma_f32_32x32x16_bf16"
# row sums l[q] = sum_k p[q][k] of the (undropped, packed) p: "dot2" = 8 v_dot2c_f32_bf16 per unit into two chains per query tile (in / out
# operands of the statement), "mfma" = two more MFMAs per unit against a fragment of ones into a second accumulator per query tile
# (a96-a159: every row of it is the row sum) -- 16 issue cycles instead of 32-68 (the dot products pay 8.5 cycles beside an MFMA)
LSUM = os.environ.get("GEN_FWD_LSUM", "dot2")   # measured (profiles/archive/r5_aa): mfma 0.4277 ms against 0.4190 for dot2 (compiled kernel 0.507):
                                                # the two extra MFMAs cost more (clock) than the eight dot products they replace

V0 = 48
SC = [48, 64]                 # score tile S^T / p (16 regs), two buffers
PK = [80, 88]                 # packed p (8 regs), two buffers
KF = [96, 104]                # K row fragments of a key tile: k-step 0 (4 regs), k-step 1 (4 regs); two sets
VT = [112, 120]               # V^T fragments: k-step 0, k-step 1; two sets
BW = [128, 136]               # the key tile's 8 pair words of the mask; two sets
XT = 144                      # the mask's 8 pair words: xor, subtract, shift in place
ONES = 152                    # bf16 (1.0, 1.0) x 4 registers (one A fragment of ones; the dot products use the first)
V_END = 156
A_QF, A_ACC = 0, 32           # a0-a31 Q fragments [qt][s] x 4, a32-a95 O^T accumulators [qt] x 16
A_L = A_ACC + 16 * QT         # a96-a159 row-sum accumulators [qt] x 16 (LSUM = "mfma")
A_END = A_L + (16 * QT if LSUM == "mfma" else 0)


def qf(qt, s): return ar(A_QF + 8 * qt + 4 * s, 4)
def acc(qt): return ar(A_ACC + 16 * qt, 16)
def lacc(qt): return ar(A_L + 16 * qt, 16)


def gen_stage(drop: bool):
    st = Stream()
    ndot = 1 if LSUM == "dot2" else 0
    nv = 16 + 8 * ((5 if drop else 1) + ndot)          # vector instructions of a unit: 64 / 56 with dropout, 32 / 24 without

    def load_tile(kt):
        b = kt & 1
        st.lds(f"ds_read_b128 {vr(KF[b], 4)}, %[a_k0] offset:{TILE * kt}", f"k{kt}")
        st.lds(f"ds_read_b128 {vr(KF[b] + 4, 4)}, %[a_k1] offset:{TILE * kt}", f"k{kt}")
        for s in range(2):
            st.lds(f"ds_read_b64_tr_b16 {vr(VT[b] + 4 * s, 2)}, %[a_v0] offset:{TILE * (KT + kt) + 1024 * s}", f"v{kt}")
            st.lds(f"ds_read_b64_tr_b16 {vr(VT[b] + 4 * s + 2, 2)}, %[a_v1] offset:{TILE * (KT + kt) + 1024 * s}", f"v{kt}")
        if drop:
            st.lds(f"ds_read_b128 {vr(BW[b], 4)}, %[a_w] offset:{64 * kt}", f"w{kt}")
            st.lds(f"ds_read_b128 {vr(BW[b] + 4, 4)}, %[a_w] offset:{64 * kt + 16}", f"w{kt}")

    def mfma_S(u, s):
        if os.environ.get("GEN_FWD_LAB", "") == "nomfma":
            return
        kt, qt, b = u // QT, u % QT, u & 1
        c = "0" if s == 0 else vr(SC[b], 16)
        st.ins(f"{MFMA} {vr(SC[b], 16)}, {vr(KF[kt & 1] + 4 * s, 4)}, {qf(qt, s)}, {c}", (f"k{kt}",) if qt == 0 and s == 0 else ())

    def mfma_PV(u, s):
        if os.environ.get("GEN_FWD_LAB", "") == "nomfma":
            return
        kt, qt, b = u // QT, u % QT, u & 1
        st.ins(f"{MFMA} {acc(qt)}, {vr(VT[kt & 1] + 4 * s, 4)}, {vr(PK[b] + 4 * s, 4)}, {acc(qt)}", (f"v{kt}",) if qt == 0 and s == 0 else ())

    def mfma_L(u, s):
        """row sums of unit u's packed p BEFORE the mask is applied: issued inside unit u, behind its packs, ahead of its and-block"""
        if os.environ.get("GEN_FWD_LAB", "") == "nomfma":
            return
        kt, qt, b = u // QT, u % QT, u & 1
        st.ins(f"{MFMA} {lacc(qt)}, {vr(ONES, 4)}, {vr(PK[b] + 4 * s, 4)}, {lacc(qt)}")

    def valu_unit(u):
        """breadth first: no instruction reads the result of the one in front of it (a dependent pair costs the vector unit a bubble:
        the depth-first order -- pack, row sum, xor, subtract, shift, and per pair -- measured 0.494 ms against 0.529 for the compiled
        kernel; this order ...)"""
        kt, qt, b = u // QT, u % QT, u & 1
        S, P, W = SC[b], PK[b], BW[kt & 1]
        seq = []
        for r in range(16):
            seq.append(lambda r=r: st.ins(f"v_exp_f32 {vr(S + r)}, {vr(S + r)}"))
        for j in range(8):
            seq.append(lambda j=j: st.ins(f"v_cvt_pk_bf16_f32 {vr(P + j)}, {vr(S + 2 * j)}, {vr(S + 2 * j + 1)}"))
        if drop:
            for j in range(8):
                nd = (f"w{kt}",) if (qt == 0 and j == 0) else ()
                seq.append(lambda j=j, nd=nd: st.ins(f"v_xor_b32 {vr(XT + j)}, {vr(W + j)}, %[aw{qt}]", nd))
                if ndot:
                    seq.append(lambda j=j: st.ins(f"v_dot2c_f32_bf16 %[l{j & 1}{qt}], {vr(P + j)}, {vr(ONES)}"))   # l stays undropped
            for j in range(8):
                seq.append(lambda j=j: st.ins(f"v_pk_sub_i16 {vr(XT + j)}, %[tpk], {vr(XT + j)} clamp"))       # < 0 iff kept
            for j in range(8):
                seq.append(lambda j=j: st.ins(f"v_pk_ashrrev_i16 {vr(XT + j)}, 15, {vr(XT + j)} op_sel_hi:[0,1]"))   # the inline 15 for BOTH halves
            for j in range(8):
                seq.append(lambda j=j: st.ins(f"v_and_b32 {vr(P + j)}, {vr(P + j)}, {vr(XT + j)}"))
        elif ndot:
            for j in range(8):
                seq.append(lambda j=j: st.ins(f"v_dot2c_f32_bf16 %[l{j & 1}{qt}], {vr(P + j)}, {vr(ONES)}"))
        assert len(seq) == nv
        return seq

    NU = KT * QT
    lab = os.environ.get("GEN_FWD_LAB", "")      # measurement builds (results invalid): nomfma = no MFMA in the loop, novalu = no vector instruction
    # positions of the unit's four MFMAs in its vector stream (index of the vector instruction each is issued in front of):
    #   S(u+1) k-steps 0, 1 between the exponentials (64 cycles apart: the second accumulates into the first; the other score buffer's
    #   last reader was the previous unit's last pack; the result is read by the next unit's first v_exp_f32, >= 20 instructions on);
    #   PV(u-1) k-step 0 in front of the packs; k-step 1 (same accumulator: a dependent MFMA issued less than ~64 cycles behind its
    #   predecessor stalls the wave -- the loop without vector instructions runs at 60 cycles per MFMA, profiles/archive/r5_ac) BEHIND the
    #   row-sum block: v_dot2c_f32_bf16 costs 8.5 instead of 4 cycles while the matrix pipe runs (tools/lab/inst_cost.hip), so that
    #   block stays clear of the MFMAs' 32 cycles.
    if LSUM == "mfma":
        # six MFMAs per unit; the packs end at 24, the and-block starts at nv - 8: L0 / L1 read the unmasked packs in between
        # (without the mask the packs stay intact: the row-sum MFMAs of unit u - 1 ride in unit u like its PV products -- six MFMAs
        # at four-instruction spacing, the unit is bound by the matrix pipe: 6 x 32 cycles)
        P = {"S0": 1, "S1": 9, "PV0": 16, "L0": 26, "L1": 34, "PV1": 43} if drop else {"S0": 0, "S1": 4, "PV0": 8, "PV1": 12, "L0": 16, "L1": 20}
    else:
        P = {"S0": 1, "S1": 9, "PV0": 16, "PV1": 40 if drop else 29}
        if os.environ.get("GEN_FWD_POS") and drop:      # measurement builds: "S0,S1,PV0,PV1"
            P = dict(zip(("S0", "S1", "PV0", "PV1"), (int(x) for x in os.environ["GEN_FWD_POS"].split(","))))
    assert P["PV1"] < nv - 1 and nv - P["S1"] >= 12 and max(P.values()) < nv
    # ---- prologue ----------------------------------------------------------------------------------------------------------------
    for i in range(4 if LSUM == "mfma" else 1):
        st.ins(f"v_mov_b32 {vr(ONES + i)}, 0x3f803f80")
    load_tile(0)
    mfma_S(0, 0)
    mfma_S(0, 1)
    load_tile(1)
    st.ins("s_nop 7")
    st.ins("s_nop 7")          # MFMA result -> VALU read: 8-pass XDL write needs >= 11 wait states
    for u in range(NU):
        kt, qt = u // QT, u % QT
        seq = valu_unit(u)
        extra = {}

        def at(pos_, fn):
            extra.setdefault(min(pos_, nv - 1), []).append(fn)
        if u > 0:
            at(P["PV0"], lambda u=u: mfma_PV(u - 1, 0))
            at(P["PV1"], lambda u=u: mfma_PV(u - 1, 1))
        if u + 1 < NU:
            at(P["S0"], lambda u=u: mfma_S(u + 1, 0))
            at(P["S1"], lambda u=u: mfma_S(u + 1, 1))
        if LSUM == "mfma" and drop:
            at(P["L0"], lambda u=u: mfma_L(u, 0))
            at(P["L1"], lambda u=u: mfma_L(u, 1))
        if LSUM == "mfma" and not drop and u > 0:
            at(P["L0"], lambda u=u: mfma_L(u - 1, 0))
            at(P["L1"], lambda u=u: mfma_L(u - 1, 1))
        if qt == 1 and kt >= 1 and kt + 1 < KT:
            # the other fragment set was last read by PV(kt-1, QT-1), issued in unit (kt, 0): free from unit (kt, 1) on
            at(P["PV1"] + 2, lambda kt=kt: load_tile(kt + 1))
        for i, fn in enumerate(seq):
            for e in extra.get(i, ()):
                e()
            if lab != "novalu":
                fn()

    u = NU - 1
    st.ins("s_nop 1")
    mfma_PV(u, 0)
    mfma_PV(u, 1)
    if LSUM == "mfma" and not drop:
        mfma_L(u, 0)
        mfma_L(u, 1)
    st.ins("s_waitcnt lgkmcnt(0)")
    st.ins("s_nop 7")
    st.ins("s_nop 7")
    st.ins("s_nop 7")          # the accumulators are read (v_accvgpr_read, compiler-scheduled) right behind the last statement
    return st.render()


CLOBBER_V = [f"v{i}" for i in range(V0, V_END)]
CLOBBER_A = [f"a{i}" for i in range(A_ACC, A_END)]


def main():
    out = []
    out.append("// GENERATED by gen_attn_fwd_asm.py -- do not edit (the Makefile rebuilds and compares it).")
    out.append(f"// {QT} query tiles per wave, {KT} key tiles per stage; asm-owned registers v{V0}-v{V_END - 1}; AGPRs a{A_QF}-a{A_ACC - 1} Q fragments (read only), a{A_ACC}-a{A_END - 1} O^T accumulators.")
    for name, drop in (("DROP", True), ("NODROP", False)):
        lines = gen_stage(drop)
        n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
        n_valu = sum(1 for ln in lines if ln.startswith("v_") and not ln.startswith("v_mfma"))
        n_lds = sum(1 for ln in lines if ln.startswith("ds_"))
        out.append(f"// {name}: {len(lines)} instructions per stage: {n_mfma} MFMA, {n_valu} vector, {n_lds} LDS")
        out.append(f"#define GAOT_ATTN_FWD_STAGE_ASM_{name} \\")
        out.append(" \\\n".join(c_string(lines).split("\n")))
        out.append("")
    zero = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(A_ACC, A_END)]
    out.append("#define GAOT_ATTN_FWD_ASM_ZERO_ACC \\")
    out.append(" \\\n".join(c_string(zero).split("\n")))
    out.append("")
    cl = ", ".join(f'"{r}"' for r in CLOBBER_V + CLOBBER_A + ["memory"])
    out.append(f"#define GAOT_ATTN_FWD_STAGE_ASM_CLOBBERS {cl}")
    out.append("#define GAOT_ATTN_FWD_ASM_ACC_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(A_ACC, A_END)) + ', "memory"')
    out.append(f"#define GAOT_ATTN_FWD_ASM_QT {QT}")
    out.append(f"#define GAOT_ATTN_FWD_ASM_KT {KT}")
    out.append(f"#define GAOT_ATTN_FWD_ASM_LSUM_MFMA {1 if LSUM == 'mfma' else 0}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
