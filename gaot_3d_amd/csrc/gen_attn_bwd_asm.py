#!/usr/bin/env python3
"""Generator of the hand-scheduled tile loop of the fused attention backward (csrc/attn_bwd_asm.inc; k_attn_bwd_asm in
attn_bf16.hip).  Reference operator: F.scaled_dot_product_attention's backward, src/model/layers/attn.py:122-127.

One workgroup = 4 waves = ONE wave per SIMD with the whole 512-register file; a wave owns 4 key blocks (128 keys) and walks the
stage's NT query tiles.  Per (query tile, key block) UNIT the arithmetic is exactly k_attn_bwd_fused's: S^T / dP^T (4 MFMAs, the
row constants -lse / -delta as the C operand), p = exp2(S'), the dropout mask on packed row words (xor + 2 SDWA compares per
query pair), P_drop / dS (4 selects, 2 multiplies, 2 cvt_pk per pair), dV^T / dK^T (4 MFMAs), the dS tile turned from "key on
the lane" to "query on the lane" (TR below), dQ^T (2 MFMAs).  What the compiler could not do with 512 registers (VERDICT r4 #1:
it moved every MFMA result through AGPR copies and spilled) is done by hand here:
  * AGPRs hold what only the matrix pipe touches: dK^T / dV^T accumulators (a0-a127), the K / V row fragments (B operands of
    S / dP: a128-a191), the K^T fragments (A operand of dQ^T: a192-a223), the permutation fragments of the transpose
    (a224-a231) -- no v_accvgpr traffic in the loop;
  * the score tiles live in VGPRs (double buffered: the 4 MFMAs of unit u + 1 are issued inside unit u's vector stream);
  * a unit's ~104 vector instructions carry its 10-12 MFMAs in their gaps (one per ~9 instructions = ~40 issue cycles >= the
    MFMA's 32): T(u-1) (transpose), B'(u-1) second halves, C(u-1) (dQ), A(u+1) (S, dP), B(u) first halves --
    MI355X_MICROARCH.md "single-issue instructions hidden per MFMA gap";
  * every LDS read is issued a unit (or a tile) ahead of its use and waited for by COUNT (s_waitcnt lgkmcnt(n), computed here).
The stage code around the tile loop (global loads, LDS staging, slot reduction, barriers) stays C++.

Measured (MI355X, S = 16 384, H = 8, profiles/archive/r5_b ... r5_e, r5_d_*mfma_t*): the loop runs at its instruction-issue time (562
cycles per unit without its LDS instructions = 104 vector instructions + 10 MFMAs to the cycle; an LDS instruction costs ~13
more on the same port), which is why the transpose moved to the matrix pipe under dropout.

Usage: python3 gen_attn_bwd_asm.py > attn_bwd_asm.inc   (tests/test_host_cpu.py checks that the committed file is current).
Environment (measurement builds only): GEN_NT, GEN_TR_DROP / GEN_TR_NODROP = mfma | lds, GEN_LAB = nowait | nolds (invalid results).
"""
import sys

import os
NT = int(os.environ.get("GEN_NT", "4"))            # query tiles per stage
W = 4             # waves per workgroup
KB = 4            # key blocks per wave
QS = 32 * NT
TILE = 2048
# how the unit's dS tile gets from "key on the lane" (the accumulator layout it is born in: B operand of dK^T) to "query on the lane"
# (B operand of dQ^T): "mfma" = TWO MORE MFMAs against a permutation ("identity") fragment -- the accumulator as the A operand
# [key][query] times I[query][query'] comes out with the query on the lane, exactly (products with 1.0 / 0.0), then 8 cvt_pk;
# "lds" = through a wave-private LDS tile (4 ds_write_b64 + 4 ds_read_b64_tr_b16 per unit).  LDS instructions cost ~13 issue
# cycles each on the port the vector stream needs (profiles/archive/r5_e): the MFMA form trades 8 of a unit's 15 for 2 MFMAs + 8 packs.
TR_DROP = os.environ.get("GEN_TR_DROP", "mfma")       # measured: 0.854 (mfma) against 0.896 ms for the compiled kernel on one box, lds 0.838 / 0.843
ORDER = os.environ.get("GEN_ORDER", "depth")      # the unit's vector stream: "depth" (pair by pair; shipped) | "breadth" (two pairs at a time, kind by kind, in
                                                   # place: measured 0.8166 / 0.8188 against 0.8185 / 0.8179 ms, profiles/archive/r5_ag -- nothing; the forward's loop gained 4 % from it)
DS_FORM = os.environ.get("GEN_DS", "fmac")           # dropout: dS = p (-delta') + (keep p) dP' (mul + fmac) | "select": p select(keep, dP' - delta', -delta') (measurement builds)
TR_NODROP = os.environ.get("GEN_TR_NODROP", "lds")   # without dropout the unit is MFMA bound: 12 MFMAs cost more than the LDS round trip (0.726 against 0.684 ms)

# ---- register map -----------------------------------------------------------------------------------------------------------
V0 = 48           # first VGPR the asm owns (the compiler keeps v0 .. V0-1 for what lives across the block)


def vr(base, n=1):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


def ar(base, n=1):
    return f"a[{base}:{base + n - 1}]" if n > 1 else f"a{base}"


SB = [48, 64]                 # S^T tile (16 regs), two buffers
DB = [80, 96]                 # dP^T tile
LC = 112                      # -lse * log2e of the tile's 32 queries in accumulator layout (C operand of S)
DC = [128, 144]               # -delta * keep, two buffers (the next tile's values arrive while the selects still read this one's)
QA0, DA0, QA1, DA1 = 160, 164, 168, 172     # Q / dO row fragments (A operands of S / dP)
DC0, QC0, DC1, QC1 = 176, 180, 184, 188     # dO / Q column fragments (A operands of dV^T / dK^T)
W8 = 192                      # 8 packed row words of the tile
PPK, DSPK = 200, 208          # packed bf16 P_drop and dS of the unit (8 regs each)
DQ = 216                      # dQ^T accumulator of the tile (16 regs)
DSF = 232                     # transposed dS fragments (B operand of dQ^T): 2 x 4 regs
XR = [240, 241]               # xor results
TMP = [[242, 243, 244, 245, 246, 247], [242, 243, 244, 245, 246, 247]]   # p0 p1 pm0 pm1 t0 t1 (in-order issue: one set serves every pair)
FREE_V = list(range(248, 254))  # left to the compiler
ADR = [254, 255]              # xor-ed LDS addresses
SM0 = 64                      # s[64:95]: the 16 lane masks of a unit (pair j: s[64+4j:65+4j], s[66+4j:67+4j])

A_DKT, A_DVT, A_KF, A_VF, A_KTF, A_ID = 0, 64, 128, 160, 192, 224
A_END = A_ID + 8


def dkt(kb): return ar(A_DKT + 16 * kb, 16)
def dvt(kb): return ar(A_DVT + 16 * kb, 16)
def kf(kb, s): return ar(A_KF + 8 * kb + 4 * s, 4)
def vf(kb, s): return ar(A_VF + 8 * kb + 4 * s, 4)
def ktf(kb, s): return ar(A_KTF + 8 * kb + 4 * s, 4)


MFMA = "v_mfma_f32_32x32x16_bf16"


class Stream:
    """program-order list of records; LDS reads carry a tag, consumers name the tags they need: the s_waitcnt lgkmcnt(n)
    in front of a consumer is computed from the number of LDS operations issued behind the youngest tag it needs"""

    def __init__(self):
        self.rec = []

    def ins(self, text, needs=()):
        self.rec.append(("ins", text, tuple(needs)))

    def lds(self, text, tag=None, needs=()):
        self.rec.append(("lds", text, tag, tuple(needs)))

    def render(self):
        lab = os.environ.get("GEN_LAB", "")      # measurement builds (results invalid): nowait = no s_waitcnt, nolds = no LDS at all
        out = []
        issued = 0            # LDS operations issued so far
        pos = {}              # tag -> index (1-based count) of the LDS operation that produces it
        waited = 0            # every operation with index <= waited is known complete
        for r in self.rec:
            needs = r[2] if r[0] == "ins" else r[3]
            need_idx = 0
            for t in needs:
                if t not in pos:
                    raise RuntimeError(f"tag {t} used before its load was issued: {r[1]}")
                need_idx = max(need_idx, pos[t])
            if need_idx > waited:
                n = issued - need_idx
                if n < 15 and not lab:     # with 15 or more younger operations outstanding the counter cannot express it: it has returned
                    out.append(f"s_waitcnt lgkmcnt({n})")
                waited = need_idx
            if r[0] == "lds":
                issued += 1
                if r[2] is not None:
                    pos[r[2]] = issued
                if lab == "nolds":
                    continue
            out.append(r[1])
        return out


def gen_stage(drop: bool):
    TR = TR_DROP if drop else TR_NODROP
    st = Stream()
    nv = 104 if drop else 48          # vector instructions of a unit

    # ---- loads ---------------------------------------------------------------------------------------------------------------
    def load_consts(t):
        for g in range(4):
            st.lds(f"ds_read_b128 {vr(LC + 4 * g, 4)}, %[a_const] offset:{128 * t + 32 * g}", f"lc{t}")
        for g in range(4):
            st.lds(f"ds_read_b128 {vr(DC[t & 1] + 4 * g, 4)}, %[a_const] offset:{4 * QS + 128 * t + 32 * g}", f"dc{t}")

    def load_rows(t):
        st.lds(f"ds_read_b128 {vr(QA0, 4)}, %[a_r0] offset:{TILE * t}", f"rows{t}")
        st.lds(f"ds_read_b128 {vr(DA0, 4)}, %[a_r0] offset:{TILE * (NT + t)}", f"rows{t}")
        st.lds(f"ds_read_b128 {vr(QA1, 4)}, %[a_r1] offset:{TILE * t}", f"rows{t}")
        st.lds(f"ds_read_b128 {vr(DA1, 4)}, %[a_r1] offset:{TILE * (NT + t)}", f"rows{t}")

    def load_cols(t):
        for (reg, tile, s) in ((DC0, NT + t, 0), (QC0, t, 0), (DC1, NT + t, 1), (QC1, t, 1)):
            st.lds(f"ds_read_b64_tr_b16 {vr(reg, 2)}, %[a_c0] offset:{TILE * tile + 1024 * s}", f"cols{t}")
            st.lds(f"ds_read_b64_tr_b16 {vr(reg + 2, 2)}, %[a_c1] offset:{TILE * tile + 1024 * s}", f"cols{t}")

    def load_w8(t):
        if drop:
            st.lds(f"ds_read_b128 {vr(W8, 4)}, %[a_w] offset:{64 * t}", f"w8{t}")
            st.lds(f"ds_read_b128 {vr(W8 + 4, 4)}, %[a_w] offset:{64 * t + 16}", f"w8{t}")

    def load_dsf(u):
        kb = u % KB
        for s in range(2):
            st.lds(f"ds_read_b64_tr_b16 {vr(DSF + 4 * s, 2)}, %[a_dc0] offset:{TILE * kb + 1024 * s}", f"dsf{u}")
            st.lds(f"ds_read_b64_tr_b16 {vr(DSF + 4 * s + 2, 2)}, %[a_dc1] offset:{TILE * kb + 1024 * s}", f"dsf{u}")

    # ---- MFMA groups ---------------------------------------------------------------------------------------------------------
    def mfma_A(u, i):
        """S^T / dP^T of unit u (issued one unit ahead): i = 0: S k-step 0, 1: dP k-step 0, 2: S k-step 1, 3: dP k-step 1"""
        t, kb, b = u // KB, u % KB, u & 1
        needs = (f"rows{t}", f"lc{t}", f"dc{t}") if kb == 0 else ()
        if i == 0:
            st.ins(f"{MFMA} {vr(SB[b], 16)}, {vr(QA0, 4)}, {kf(kb, 0)}, {vr(LC, 16)}", needs)
        elif i == 1:
            # with dropout the row constants -delta' enter as a multiplicand of p instead (valu_unit): the product starts from 0
            st.ins(f"{MFMA} {vr(DB[b], 16)}, {vr(DA0, 4)}, {vf(kb, 0)}, {'0' if (drop and DS_FORM != 'select') else vr(DC[t & 1], 16)}", needs)
        elif i == 2:
            st.ins(f"{MFMA} {vr(SB[b], 16)}, {vr(QA1, 4)}, {kf(kb, 1)}, {vr(SB[b], 16)}")
        else:
            st.ins(f"{MFMA} {vr(DB[b], 16)}, {vr(DA1, 4)}, {vf(kb, 1)}, {vr(DB[b], 16)}")

    def mfma_B(u, i):
        """dV^T / dK^T of unit u: i = 0: dV k-step 0 (P pairs 0-3), 1: dK k-step 0, 2: dV k-step 1, 3: dK k-step 1"""
        t, kb = u // KB, u % KB
        needs = (f"cols{t}",) if (kb == 0 and i == 0) else ()
        if i == 0:
            st.ins(f"{MFMA} {dvt(kb)}, {vr(DC0, 4)}, {vr(PPK, 4)}, {dvt(kb)}", needs)
        elif i == 1:
            st.ins(f"{MFMA} {dkt(kb)}, {vr(QC0, 4)}, {vr(DSPK, 4)}, {dkt(kb)}")
        elif i == 2:
            st.ins(f"{MFMA} {dvt(kb)}, {vr(DC1, 4)}, {vr(PPK + 4, 4)}, {dvt(kb)}")
        else:
            st.ins(f"{MFMA} {dkt(kb)}, {vr(QC1, 4)}, {vr(DSPK + 4, 4)}, {dkt(kb)}")

    def mfma_C(u, s):
        """dQ^T of the tile += K^T(kb) dS^T(u), k-step s"""
        kb = u % KB
        c = "0" if (kb == 0 and s == 0) else vr(DQ, 16)
        st.ins(f"{MFMA} {vr(DQ, 16)}, {ktf(kb, s)}, {vr(DSF + 4 * s, 4)}, {c}", (f"dsf{u}",) if TR == "lds" else ())

    def mfma_T(u, s):
        """dS^T(u) with the query on the lane = dS-as-A-operand x permutation fragment, k-step s, into unit u's (dead) S buffer"""
        b = u & 1
        c = "0" if s == 0 else vr(SB[b], 16)
        st.ins(f"{MFMA} {vr(SB[b], 16)}, {vr(DSPK + 4 * s, 4)}, {ar(A_ID + 4 * s, 4)}, {c}")

    def cvt_T(u, j):
        b = u & 1
        st.ins(f"v_cvt_pk_bf16_f32 {vr(DSF + j)}, {vr(SB[b] + 2 * j)}, {vr(SB[b] + 2 * j + 1)}")

    def store_ds(u, half):
        """the unit's packed dS, k-step `half`: two 8-byte stores into the wave's dS tile [key][query]"""
        kb = u % KB
        for c in (2 * half, 2 * half + 1):
            if c == 0:
                adr = "%[a_ds]"
            else:
                st.ins(f"v_xor_b32 {vr(ADR[c & 1])}, {16 * c}, %[a_ds]")
                adr = vr(ADR[c & 1])
            st.lds(f"ds_write_b64 {adr}, {vr(DSPK + 2 * c, 2)} offset:{TILE * kb}")

    def store_slot(t):
        off, base = t * W * 4096, "%[a_slot]"
        if off >= 65536:            # beyond the 16-bit offset field: one add (XR is free outside the compare phase)
            st.ins(f"v_add_u32 {vr(XR[0])}, 0x10000, %[a_slot]")
            off, base = off - 65536, vr(XR[0])
        for g in range(4):
            if g == 0:
                adr = base
            else:
                st.ins(f"v_xor_b32 {vr(ADR[g & 1])}, {32 * g}, {base}")
                adr = vr(ADR[g & 1])
            st.lds(f"ds_write_b128 {adr}, {vr(DQ + 4 * g, 4)} offset:{off}")

    # ---- the vector stream of a unit: a list of closures, one per instruction --------------------------------------------------
    def valu_unit(u):
        """ORDER (GEN_ORDER): "depth" = pair by pair (exp, exp, select, select, multiply, multiply-add, pack, pack: every instruction
        reads a result one or two instructions old); "breadth" = two pairs at a time, kind by kind, everything in place -- p in the score
        registers, dS in the dP registers (dP <- (keep p) dP', then += p (-delta')), only the kept p in temporaries -- so that no
        instruction reads a result younger than four instructions (the forward's tile loop gained 4 % from this order)"""
        t, kb, b = u // KB, u % KB, u & 1
        S, DP, DCt = SB[b], DB[b], DC[t & 1]
        seq = []
        def M(j, h): return f"s[{SM0 + 4 * j + 2 * h}:{SM0 + 4 * j + 2 * h + 1}]"
        if drop:
            if ORDER == "breadth":
                for j in range(0, 8, 2):
                    nd = (f"w8{t}",) if (j == 0) else ()
                    seq.append(lambda j=j, nd=nd: st.ins(f"v_xor_b32 {vr(XR[0])}, {vr(W8 + j)}, %[bsel{kb}]", nd))
                    seq.append(lambda j=j: st.ins(f"v_xor_b32 {vr(XR[1])}, {vr(W8 + j + 1)}, %[bsel{kb}]"))
                    for h in range(2):
                        for jj in range(2):
                            seq.append(lambda j=j, h=h, jj=jj: st.ins(f"v_cmp_ge_u32_sdwa {M(j + jj, h)}, {vr(XR[jj])}, %[thr] src0_sel:WORD_{h} src1_sel:DWORD"))
            else:
                for j in range(8):
                    x = XR[j & 1]
                    nd = (f"w8{t}",) if (j == 0) else ()
                    seq.append(lambda x=x, j=j, nd=nd: st.ins(f"v_xor_b32 {vr(x)}, {vr(W8 + j)}, %[bsel{kb}]", nd))
                    seq.append(lambda x=x, j=j: st.ins(f"v_cmp_ge_u32_sdwa {M(j, 0)}, {vr(x)}, %[thr] src0_sel:WORD_0 src1_sel:DWORD"))
                    seq.append(lambda x=x, j=j: st.ins(f"v_cmp_ge_u32_sdwa {M(j, 1)}, {vr(x)}, %[thr] src0_sel:WORD_1 src1_sel:DWORD"))
        if ORDER == "breadth":
            T4 = TMP[0][:4]
            for j in range(0, 8, 2):
                rs = [2 * j, 2 * j + 1, 2 * j + 2, 2 * j + 3]
                for r in rs:
                    seq.append(lambda r=r: st.ins(f"v_exp_f32 {vr(S + r)}, {vr(S + r)}"))
                if drop:
                    for i, r in enumerate(rs):
                        seq.append(lambda i=i, r=r: st.ins(f"v_cndmask_b32_e64 {vr(T4[i])}, 0, {vr(S + r)}, {M(r // 2, r & 1)}"))
                    for i, r in enumerate(rs):
                        seq.append(lambda i=i, r=r: st.ins(f"v_mul_f32 {vr(DP + r)}, {vr(T4[i])}, {vr(DP + r)}"))
                    for r in rs:
                        seq.append(lambda r=r: st.ins(f"v_fmac_f32 {vr(DP + r)}, {vr(S + r)}, {vr(DCt + r)}"))
                    seq.append(lambda j=j: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j)}, {vr(T4[0])}, {vr(T4[1])}"))
                    seq.append(lambda j=j: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j + 1)}, {vr(T4[2])}, {vr(T4[3])}"))
                else:
                    for r in rs:
                        seq.append(lambda r=r: st.ins(f"v_mul_f32 {vr(DP + r)}, {vr(S + r)}, {vr(DP + r)}"))
                    seq.append(lambda j=j, rs=rs: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j)}, {vr(S + rs[0])}, {vr(S + rs[1])}"))
                    seq.append(lambda j=j, rs=rs: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j + 1)}, {vr(S + rs[2])}, {vr(S + rs[3])}"))
                seq.append(lambda j=j, rs=rs: st.ins(f"v_cvt_pk_bf16_f32 {vr(DSPK + j)}, {vr(DP + rs[0])}, {vr(DP + rs[1])}"))
                seq.append(lambda j=j, rs=rs: st.ins(f"v_cvt_pk_bf16_f32 {vr(DSPK + j + 1)}, {vr(DP + rs[2])}, {vr(DP + rs[3])}"))
            assert len(seq) == nv, (len(seq), nv)
            return seq
        for j in range(8):
            r0, r1 = 2 * j, 2 * j + 1
            p0, p1, pm0, pm1, t0, t1 = TMP[j & 1]
            m0, m1 = M(j, 0), M(j, 1)
            seq.append(lambda p0=p0, r0=r0: st.ins(f"v_exp_f32 {vr(p0)}, {vr(S + r0)}"))
            seq.append(lambda p1=p1, r1=r1: st.ins(f"v_exp_f32 {vr(p1)}, {vr(S + r1)}"))
            if drop:
                # dS = p (keep dP' - delta') = p (-delta') + (keep p) dP': the dP MFMA starts from C = 0 (DROP_FMAC), the kept p the dV product
                # needs anyway carries the mask, and the two-operand v_mul_f32 / v_fmac_f32 cost 4 issue cycles where a second
                # v_cndmask_b32_e64 costs 5 (tools/lab/inst_cost.hip): 13 instead of 14 cycles per element
                seq.append(lambda pm0=pm0, p0=p0, m0=m0: st.ins(f"v_cndmask_b32_e64 {vr(pm0)}, 0, {vr(p0)}, {m0}"))
                if DS_FORM == "select":
                    seq.append(lambda t0=t0, r0=r0, m0=m0: st.ins(f"v_cndmask_b32_e64 {vr(t0)}, {vr(DCt + r0)}, {vr(DP + r0)}, {m0}"))
                seq.append(lambda pm1=pm1, p1=p1, m1=m1: st.ins(f"v_cndmask_b32_e64 {vr(pm1)}, 0, {vr(p1)}, {m1}"))
                if DS_FORM == "select":
                    seq.append(lambda t1=t1, r1=r1, m1=m1: st.ins(f"v_cndmask_b32_e64 {vr(t1)}, {vr(DCt + r1)}, {vr(DP + r1)}, {m1}"))
                    seq.append(lambda t0=t0, p0=p0: st.ins(f"v_mul_f32 {vr(t0)}, {vr(p0)}, {vr(t0)}"))
                    seq.append(lambda t1=t1, p1=p1: st.ins(f"v_mul_f32 {vr(t1)}, {vr(p1)}, {vr(t1)}"))
                else:
                    seq.append(lambda t0=t0, p0=p0, r0=r0: st.ins(f"v_mul_f32 {vr(t0)}, {vr(p0)}, {vr(DCt + r0)}"))
                    seq.append(lambda t1=t1, p1=p1, r1=r1: st.ins(f"v_mul_f32 {vr(t1)}, {vr(p1)}, {vr(DCt + r1)}"))
                    seq.append(lambda t0=t0, pm0=pm0, r0=r0: st.ins(f"v_fmac_f32 {vr(t0)}, {vr(pm0)}, {vr(DP + r0)}"))
                    seq.append(lambda t1=t1, pm1=pm1, r1=r1: st.ins(f"v_fmac_f32 {vr(t1)}, {vr(pm1)}, {vr(DP + r1)}"))
                seq.append(lambda j=j, pm0=pm0, pm1=pm1: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j)}, {vr(pm0)}, {vr(pm1)}"))
            else:
                seq.append(lambda t0=t0, p0=p0, r0=r0: st.ins(f"v_mul_f32 {vr(t0)}, {vr(p0)}, {vr(DP + r0)}"))
                seq.append(lambda t1=t1, p1=p1, r1=r1: st.ins(f"v_mul_f32 {vr(t1)}, {vr(p1)}, {vr(DP + r1)}"))
                seq.append(lambda j=j, p0=p0, p1=p1: st.ins(f"v_cvt_pk_bf16_f32 {vr(PPK + j)}, {vr(p0)}, {vr(p1)}"))
            seq.append(lambda j=j, t0=t0, t1=t1: st.ins(f"v_cvt_pk_bf16_f32 {vr(DSPK + j)}, {vr(t0)}, {vr(t1)}"))
        assert len(seq) == nv, (len(seq), nv)
        return seq

    NU = NT * KB
    ph1 = 24 if drop else 0                      # xor / compare instructions in front of the pair arithmetic
    per_pair = (nv - ph1) // 8
    half_done = ph1 + 4 * per_pair               # pairs 0-3 packed
    first_ds = ph1 + per_pair - 1                # the instruction that overwrites DSPK[0] (the previous unit's, read by T' k-step 0)
    nm = 12 if TR == "mfma" else 10              # MFMAs of a unit
    step = nv / float(nm)
    # positions (index of the vector instruction an item is issued in front of), in issue order:
    #   mfma: T'(u-1) k-steps 0, 1 | B'(u-1) k-step 1 (dV, dK) | C(u-1) k-steps 0, 1 | A(u+1) x 4 | B(u) k-step 0 (dV, dK)
    #   lds:                         B'(u-1)                    | C(u-1)              | A(u+1)     | B(u)
    names = (["T0", "T1"] if TR == "mfma" else []) + ["Bp2", "Bp3", "C0", "C1", "A0", "A1", "A2", "A3", "B0", "B1"]
    pos, p, cvt_at = [], 0, None
    for i, nme in enumerate(names):
        if nme == "C0" and TR == "mfma":
            # the 8 packs of the transposed tile ride as extra instructions in front of vector instructions cvt_at .. cvt_at + 7:
            # >= 12 instructions behind T1 (MFMA result -> VALU read), C0 two more behind the last of them (VALU write -> MFMA read)
            cvt_at = pos[1] + 12
            p = max(p, cvt_at + 10)
        if nme == "B0":
            p = max(p, half_done + 3)
        p = min(p, nv - 1 - 2 * (len(names) - 1 - i))      # leave two instructions per MFMA that still has to follow
        pos.append(p)
        p += max(2, (nv - 6 - p) // max(1, len(names) - 1 - i))     # (the last MFMA a few instructions before the unit's end)
    P = dict(zip(names, pos))
    assert all(pos[i] < pos[i + 1] for i in range(len(pos) - 1)) and pos[-1] < nv, P
    if TR == "mfma":
        assert P["T0"] < first_ds and P["T1"] < first_ds + 4 * per_pair and P["C0"] >= cvt_at + 10 and P["A0"] > P["C1"], (P, first_ds, cvt_at)
    assert P["Bp3"] < first_ds + 4 * per_pair and P["B0"] >= half_done + 3, P

    # ---- prologue --------------------------------------------------------------------------------------------------------------
    load_rows(0)
    load_consts(0)
    load_w8(0)
    load_cols(0)
    for i in range(4):
        mfma_A(0, i)
    if not drop:               # no compare phase in front of the first exp: MFMA result -> VALU read needs >= 11 wait states
        st.ins("s_nop 7")
        st.ins("s_nop 7")
    # ---- units -----------------------------------------------------------------------------------------------------------------
    for u in range(NU):
        t, kb = u // KB, u % KB
        seq = valu_unit(u)
        extra = {}                                # position -> list of closures issued in front of that instruction

        def at(pos_, fn):
            extra.setdefault(min(pos_, nv - 1), []).append(fn)
        if u > 0:
            if TR == "mfma":
                at(P["T0"], lambda u=u: mfma_T(u - 1, 0))
                at(P["T1"], lambda u=u: mfma_T(u - 1, 1))
                for j in range(8):
                    at(cvt_at + j, lambda u=u, j=j: cvt_T(u - 1, j))
            else:
                at(0, lambda u=u: load_dsf(u - 1))                  # behind the last dS store of unit u-1 (program order)
            at(P["Bp2"], lambda u=u: mfma_B(u - 1, 2))
            at(P["Bp3"], lambda u=u: mfma_B(u - 1, 3))
            at(P["C0"], lambda u=u: mfma_C(u - 1, 0))
            at(P["C1"], lambda u=u: mfma_C(u - 1, 1))
            if kb == 0:                                             # the previous tile's dQ^T is complete: to its slot
                at(P["C1"] + 14, lambda t=t: store_slot(t - 1))
        if kb == 0 and u > 0:
            at(P["Bp3"] + 2, lambda t=t: load_cols(t))              # behind B'(t-1, 3): the last reader of the old columns
        if u + 1 < NU:
            for i in range(4):
                at(P[f"A{i}"], lambda u=u, i=i: mfma_A(u + 1, i))
        if kb == KB - 2 and t + 1 < NT:                             # behind A(t, 3): rows / row constants of the next tile
            at(P["A3"] + 3, lambda t=t: load_rows(t + 1))
            at(P["A3"] + 5, lambda t=t: load_consts(t + 1))
        if kb == KB - 1 and t + 1 < NT and drop:                    # behind the unit's last xor: the next tile's row words
            at(ph1 + 1, lambda t=t: load_w8(t + 1))
        at(P["B0"], lambda u=u: mfma_B(u, 0))
        if TR == "lds":
            at(P["B0"] + 1, lambda u=u: store_ds(u, 0))
        at(P["B1"], lambda u=u: mfma_B(u, 1))
        for i, fn in enumerate(seq):
            for e in extra.get(i, ()):
                e()
            fn()
        if TR == "lds":
            store_ds(u, 1)
    # ---- epilogue: the last unit's second halves, its dQ^T, the last tile's slot -------------------------------------------------
    u = NU - 1
    if TR == "mfma":
        mfma_T(u, 0)
        mfma_T(u, 1)
        mfma_B(u, 2)
        mfma_B(u, 3)
        st.ins("s_nop 7")
        st.ins("s_nop 7")      # MFMA result -> VALU read: 8-pass XDL write needs >= 11 wait states
        for j in range(8):
            cvt_T(u, j)
        st.ins("s_nop 1")
    else:
        load_dsf(u)
        mfma_B(u, 2)
        mfma_B(u, 3)
    mfma_C(u, 0)
    mfma_C(u, 1)
    st.ins("s_nop 7")
    st.ins("s_nop 7")          # MFMA result -> LDS store data: 8-pass XDL write needs >= 11 wait states (CDNA3 ISA 4.5)
    store_slot(NT - 1)
    st.ins("s_waitcnt lgkmcnt(0)")
    return st.render()


CLOBBER_V = [f"v{i}" for i in range(V0, 256) if i not in FREE_V]
CLOBBER_S = [f"s{i}" for i in range(SM0, SM0 + 32)]
CLOBBER_A = [f"a{i}" for i in range(0, A_END)]


def c_string(lines):
    return "\n".join(f'    "{ln}\\n\\t"' for ln in lines)


def main():
    out = []
    out.append("// GENERATED by gen_attn_bwd_asm.py -- do not edit (the Makefile rebuilds and compares it).")
    out.append(f"// NT = {NT} query tiles per stage, {W} waves x {KB} key blocks; asm-owned registers v{V0}-v255, s{SM0}-s{SM0 + 31}, a0-a{A_END - 1}; dS transpose: {TR_DROP} (dropout) / {TR_NODROP} (none).")
    for name, drop in (("DROP", True), ("NODROP", False)):
        lines = gen_stage(drop)
        n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
        n_valu = sum(1 for ln in lines if ln.startswith("v_") and not ln.startswith("v_mfma"))
        n_lds = sum(1 for ln in lines if ln.startswith("ds_"))
        out.append(f"// {name}: {len(lines)} instructions per stage: {n_mfma} MFMA, {n_valu} vector, {n_lds} LDS")
        out.append(f"#define GAOT_ATTN_BWD_STAGE_ASM_{name} \\")
        body = c_string(lines).split("\n")
        out.append(" \\\n".join(body))
        out.append("")
    zero = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(A_DVT + 16 * KB)]
    out.append("#define GAOT_ATTN_BWD_ASM_ZERO_ACC \\")
    out.append(" \\\n".join(c_string(zero).split("\n")))
    out.append("")
    out.append("#define GAOT_ATTN_BWD_ASM_ACC_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(A_DVT + 16 * KB)) + ', "memory"')
    cl = ", ".join(f'"{r}"' for r in CLOBBER_V + CLOBBER_S + CLOBBER_A + ["vcc", "memory"])
    out.append(f"#define GAOT_ATTN_BWD_STAGE_ASM_CLOBBERS {cl}")
    out.append(f"#define GAOT_ATTN_BWD_ASM_NT {NT}")
    out.append(f"#define GAOT_ATTN_BWD_ASM_MFMA_T_DROP {1 if TR_DROP == 'mfma' else 0}")
    out.append(f"#define GAOT_ATTN_BWD_ASM_MFMA_T_NODROP {1 if TR_NODROP == 'mfma' else 0}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
