#!/usr/bin/env python3
"""Generates engine_asm.inc: the k-loop of the MFMA contraction engine (mfma_gemm.hip) as gfx950 assembly.

Why assembly: measured on MI355X (tools/mfma_stage.hip) a two-wavefront-per-SIMD stream of v_mfma_f64_16x16x4_f64
runs at 77.8 TFLOP/s with LDS fragment reads, global loads and barriers beside it, but
  * refilling LDS through registers (global_load -> ds_write_b128) costs 8 % -- the VGPR->LDS data transfer of the
    stores stalls the matrix pipe; LDS-DMA (global_load_lds_dwordx4) costs 4 %;
  * every VALU instruction issued beside the stream costs its issue cycles (8 per k-step: 2 %, 32: 9 %) -- the
    compiled loop carried ~0.6-1.0 VALU instructions per MFMA (address updates, exec-mask bookkeeping).
The loop below has no VALU work besides the MFMAs (plain variants), refills LDS by DMA, reads fragments one k-step
ahead, and has one barrier per k-stage placed in front of the stage's last k-step.

Register map (fixed; the HIP part of the kernel is capped at 88 (fp64) / 80 (fp32) VGPRs and 84 SGPRs with
amdgpu_num_vgpr / amdgpu_num_sgpr, so the registers above are ours):
  s[84:85] / s[86:87] operand base pointers of the next stage to fetch, s[88:89] / s[90:91] weight pointers,
  s92 saved m0, s[94:95] saved exec
  v[128:255]  accumulators   fp64: sub-tile (i, j) -> v[128 + 8(4i+j) .. +7]     fp32: v[128 + 4(4i+j) .. +3]
  v[96:127]   fragments      set s: fp64 A_i = v[96+16s+2i..+1], B_j = v[96+16s+8+2j..+1]; fp32 A_i = v[96+8s+i], B_j = v[96+8s+4+j]
  v[88:95] (fp64) / v[80:95] (fp32)  x-major fragment address variants (one per k-step after the first), or the
              weight fragments (w, then w2 four registers on)

LDS map (bytes; one workgroup = 73728): A image of stage buffer b at b*18432, B image at 36864 + b*18432.
  x-major image (operand rows contiguous along k, 128 B per row): 16 chunks of 1024 B (8 rows each, one DMA
      instruction per chunk); 16-byte vector c of row r sits at (r>>3)*1024 + ((r&7)*8 + (c ^ ((r>>1)&7)))*16 --
      the XOR makes the fragment reads (16 rows x one k-pair per half-wave) conflict-free without padding, which
      LDS-DMA cannot produce (its destination is lane-linear);
  k-major image: fp64: k-row k (1024 B, one DMA instruction) at k*1152; fp32: one DMA instruction carries k-rows
      4q+e and 4q+e+2 (512 B each) to chunk 2q+e at (2q+e)*1088, so k and k+1 differ by 64 B mod 128 B in bank space;
  per-k weights of the stage (128 B) at +18304 of the A image (w) and of the B image (w2, column-sum weights).
"""
import os
import sys

# engine lab (timing only, results wrong): GEN_NOBARRIER=1 leaves the per-stage s_barrier out of the generated loops --
# what the workgroup barrier itself costs (tools/lab16.sh)
NOBARRIER = os.environ.get("GEN_NOBARRIER") == "1"

OPSZ = 18432           # bytes per operand image
BUFSZ = OPSZ           # stage buffer b of an operand sits at b*OPSZ from the operand's base
WOFF = 18304           # weights of the stage inside an operand image


class Cfg:
    def __init__(self, f64, op, var, diag=None, cont=False, desc=False):
        self.f64, self.op, self.var, self.diag = f64, op, var, diag
        self.cont = cont                    # second phase of a two-phase item: the accumulators are kept, not zeroed
        # the k-range is walked from its last stage down to its first (the operand bases step backwards): the second,
        # short item of a column-tile pair of a triangular NN product then reads the A blocks its seven neighbours on
        # the XCD read at the same time (mfma_gemm.hip: tile_of_block).  diag "khid": the diagonal block of B, which
        # ends the k-range, is visited first, its stages in reverse order.
        # diag "klod" (NT): the diagonal block of B, which starts the k-range, is visited last (first phase of the long
        # item of a two-phase pair: the eight long items of a row panel then start together at the top of k)
        self.desc = desc or diag in ("khid", "klod")
        self.nkk = 4 if f64 else 8          # k-steps (of 4) per stage
        self.fw = 2 if f64 else 1           # dwords per fragment element
        self.aw = 8 if f64 else 4           # accumulator registers per 16x16 sub-tile
        self.a_x = op in ("nn", "nt")       # A operand x-major?
        self.b_x = op == "nt"
        self.rd = "ds_read_b64" if f64 else "ds_read_b32"
        self.mfma = "v_mfma_f64_16x16x4_f64" if f64 else "v_mfma_f32_16x16x4_f32"

    def name(self):
        return "ENGINE_LOOP_%s_%s%s%s%s" % ("F64" if self.f64 else "F32", self.op.upper(),
                                            {0: "", 1: "_W", 2: "_WS"}[self.var],
                                            {None: "", "khi": "_KHI", "klo": "_KLO", "sy": "_SY", "khid": "_KHID", "klod": "_KLOD"}[self.diag],
                                            ("_CONT" if self.cont else "") + ("_DESC" if self.desc and self.diag not in ("khid", "klod") else ""))


def reg(base, width):
    return "v%d" % base if width == 1 else "v[%d:%d]" % (base, base + width - 1)


def gen(c):
    L = []
    emit = L.append
    fw, aw, nkk = c.fw, c.aw, c.nkk

    def A(s, i):
        return reg(96 + 8 * fw * s + fw * i, fw)

    def B(s, j):
        return reg(96 + 8 * fw * s + 4 * fw + fw * j, fw)

    def ACC(i, j):
        return reg(128 + aw * (4 * i + j), aw)

    tb = 88 if c.f64 else 80                   # first of our temporaries (the HIP code is capped below it)
    # weight fragment of set s (fp32: on even registers, so that it can be the 64-bit second source of v_pk_mul_f32)
    W = lambda s: reg(tb + 2 * s, fw)
    W2 = lambda s: reg(tb + 4 + fw * s, fw)
    AX = lambda kk: "v%d" % (tb + kk - 1)            # kk >= 1 (x-major A address of k-step kk)
    BX = lambda kk: "v%d" % (tb + (nkk - 1) + kk - 1)

    # ---- fragment reads of k-step kk from stage buffer b into register set s
    def reads(b, kk, s):
        for i in range(4):
            if c.a_x:
                addr = "%[addrA]" if kk == 0 else AX(kk)
                off = b * BUFSZ + i * 2048
            else:
                addr = "%[addrA]"   # k-major A (TN): sub-tile i = 16-row block wr + 2i of the tile, like B's columns
                off = b * BUFSZ + (kk * 4608 + i * 256 if c.f64 else kk * 2176 + i * 128)
            emit("%s %s, %s offset:%d" % (c.rd, A(s, i), addr, off))
        for j in range(4):
            if c.b_x:
                addr = "%[addrB]" if kk == 0 else BX(kk)
                off = b * BUFSZ + j * 4096
            else:
                addr = "%[addrB]"
                off = b * BUFSZ + (kk * 4608 + j * 256 if c.f64 else kk * 2176 + j * 128)
            emit("%s %s, %s offset:%d" % (c.rd, B(s, j), addr, off))
        if c.var >= 1:
            emit("%s %s, %%[addrW] offset:%d" % (c.rd, W(s), b * BUFSZ + WOFF + kk * 4 * (8 if c.f64 else 4)))
        if c.var == 2:
            emit("%s %s, %%[addrW2] offset:%d" % (c.rd, W2(s), b * BUFSZ + WOFF + kk * 4 * (8 if c.f64 else 4)))

    nreads = 8 + (1 if c.var >= 1 else 0) + (1 if c.var == 2 else 0)

    # ---- the 16 MFMAs of a k-step on register set s (weighted variants scale the A fragments first)
    def mfmas(s, live=(0, 1, 2, 3)):
        if c.var == 2:   # column sums of the raw operand: csum_i += A_i * w2   (before the scaling)
            for i in range(4):
                if c.f64:
                    emit("v_fma_f64 %%[cs%d], %s, %s, %%[cs%d]" % (i, A(s, i), W2(s), i))
                else:     # float fragment, double accumulator
                    emit("v_mul_f32 %%[t0], %s, %s" % (A(s, i), W2(s)))
                    emit("v_cvt_f64_f32 %[t64], %[t0]")
                    emit("v_add_f64 %%[cs%d], %%[cs%d], %%[t64]" % (i, i))
        if c.var >= 1:
            if c.f64:
                for i in range(4):
                    emit("v_mul_f64 %s, %s, %s" % (A(s, i), A(s, i), W(s)))
            else:
                # fp32: the four A fragments of a k-step are two aligned register pairs -- two packed multiplies (the
                # weight's low half for both halves) instead of four scalar ones: every VALU instruction beside the MFMA
                # stream costs its issue cycles, and an fp32 k-step is only 512 cycles of MFMAs
                for i in (0, 2):
                    pair = "v[%d:%d]" % (96 + 8 * s + i, 96 + 8 * s + i + 1)
                    wp = "v[%d:%d]" % (tb + 2 * s, tb + 2 * s + 1)
                    emit("v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]" % (pair, pair, wp))
            emit("s_nop 1")
        for i in range(4):
            for j in range(4):
                if j in live or (i, j) in live:
                    emit("%s %s, %s, %s, %s" % (c.mfma, ACC(i, j), A(s, i), B(s, j), ACC(i, j)))

    # ---- LDS-DMA of one stage into buffer b (8 tile instructions per wavefront; weights by wavefronts 0 / 1)
    def dma(b):
        unit_a = 1024 if c.a_x else (1152 if c.f64 else 1088)
        unit_b = 1024 if c.b_x else (1152 if c.f64 else 1088)
        for p in range(4):
            emit("s_add_u32 m0, %%[m0A], %d" % (b * BUFSZ + p * unit_a))
            emit("s_nop 0")
            emit("global_load_lds_dwordx4 %%[voffA%d], s[84:85]" % p)
        for p in range(4):
            emit("s_add_u32 m0, %%[m0B], %d" % (b * BUFSZ + p * unit_b))
            emit("s_nop 0")
            emit("global_load_lds_dwordx4 %%[voffB%d], s[86:87]" % p)
        if c.var >= 1:
            # 128 bytes of weights: lanes 0..7 of wavefront 0 (w) and of wavefront 1 (w2)
            emit("s_cmp_lg_u32 %[wave], 0")
            emit("s_cbranch_scc1 3f")
            emit("s_mov_b64 s[94:95], exec")
            emit("s_mov_b64 exec, 0xff")
            emit("s_mov_b32 m0, %d" % (b * BUFSZ + WOFF))
            emit("s_nop 0")
            emit("global_load_lds_dwordx4 %[voffW], s[88:89]")
            emit("s_mov_b64 exec, s[94:95]")
            emit("3:")
            emit("s_add_u32 s88, s88, 128")
            emit("s_addc_u32 s89, s89, 0")
        if c.var == 2:
            emit("s_cmp_lg_u32 %[wave], 1")
            emit("s_cbranch_scc1 4f")
            emit("s_mov_b64 s[94:95], exec")
            emit("s_mov_b64 exec, 0xff")
            emit("s_mov_b32 m0, %d" % (2 * OPSZ + b * BUFSZ + WOFF))
            emit("s_nop 0")
            emit("global_load_lds_dwordx4 %[voffW], s[90:91]")
            emit("s_mov_b64 exec, s[94:95]")
            emit("4:")
            emit("s_add_u32 s90, s90, 128")
            emit("s_addc_u32 s91, s91, 0")
        # advance the operand bases to the next stage (descending walks: the previous one)
        if c.desc:
            emit("s_sub_u32 s84, s84, %[stepA]")
            emit("s_subb_u32 s85, s85, 0")
            emit("s_sub_u32 s86, s86, %[stepB]")
            emit("s_subb_u32 s87, s87, 0")
        else:
            emit("s_add_u32 s84, s84, %[stepA]")
            emit("s_addc_u32 s85, s85, 0")
            emit("s_add_u32 s86, s86, %[stepB]")
            emit("s_addc_u32 s87, s87, 0")

    # ---- one k-stage out of buffer b
    def stage(b, exit_label, live=None, tail=None):
        lv = (lambda kk: (0, 1, 2, 3)) if live is None else live
        for kk in range(nkk - 1):
            reads(b, kk + 1, (kk + 1) & 1)
            emit("s_waitcnt lgkmcnt(%d)" % nreads)
            mfmas(kk & 1, lv(kk))
        # last k-step: everything this wave read from buffer b has arrived; its DMA into the other buffer has landed
        emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
        if not NOBARRIER:
            emit("s_barrier")
        emit("s_cmp_lt_u32 %[rem], 3")          # a stage after the next one? -> refill this buffer
        emit("s_cbranch_scc1 5f")
        dma(b)
        emit("5:")
        emit("s_cmp_lt_u32 %[rem], 2")          # a next stage? -> its first fragments
        emit("s_cbranch_scc1 6f")
        reads(b ^ 1, 0, 0)
        emit("6:")
        mfmas((nkk - 1) & 1, lv(nkk - 1))
        emit("s_sub_u32 %[rem], %[rem], 1")
        if tail is not None:
            tail()
        elif exit_label:
            emit("s_cmp_eq_u32 %[rem], 0")
            emit("s_cbranch_scc1 %s" % exit_label)

    # ---- the diagonal 128-block of a triangular B operand (c.diag): k-step by k-step only the 16-column sub-tiles
    # that meet non-zeros are multiplied.  Sub-tile cj = wc + 2j of wave column wc; k-step at offset kpos (of 4 k)
    # inside the block:  "khi" (B[k][j] = 0 for k > j, the block ends the k-range):   live iff kpos <= 16 cj + 15
    #                    "klo" (B[j][k] = 0 for k < j, the block starts the k-range): live iff kpos + 3 >= 16 cj
    DSTAGES = 128 // (4 * nkk)

    def diag_live(u, wc):
        def f(kk):
            kpos = 4 * (u * nkk + kk)
            if c.diag in ("khi", "khid"):
                return tuple(j for j in range(4) if kpos <= 16 * (wc + 2 * j) + 15)
            # ("klo", "klod")
            return tuple(j for j in range(4) if kpos + 3 >= 16 * (wc + 2 * j))
        return f

    def diag_section(b0, wc, end_label):
        for u in range(DSTAGES):
            last = u == DSTAGES - 1
            # "khid" / "klod": the v-th stage visited is stage DSTAGES-1-v of the block
            ub = DSTAGES - 1 - u if c.diag in ("khid", "klod") else u
            stage((b0 + u) & 1, end_label if (last or c.diag in ("klo", "khid")) else None, live=diag_live(ub, wc))

    # ================= program: two asm statements per output tile =================
    # PRO : fetch of the tile's first k-stage (issued by the HIP code before the previous tile's epilogue, so the
    #       fetch latency hides under the epilogue's stores)
    # MAIN: zero the accumulators, publish stage 0, run the k-loop
    emit("s_mov_b32 s92, m0")
    emit("s_mov_b64 s[84:85], %[baseA]")
    emit("s_mov_b64 s[86:87], %[baseB]")
    if c.var >= 1:
        emit("s_mov_b64 s[88:89], %[baseW]")
    if c.var == 2:
        emit("s_mov_b64 s[90:91], %[baseW2]")
    dma(0)
    emit("s_mov_b32 m0, s92")
    pro, L[:] = list(L), []

    emit("s_mov_b32 s92, m0")
    # x-major fragment address variants: address(kk) = address(0) ^ (kk * 32 bytes [fp64] / 16 bytes [fp32])
    if c.a_x:
        for kk in range(1, nkk):
            emit("v_xor_b32 %s, %d, %%[addrA]" % (AX(kk), kk * (32 if c.f64 else 16)))
    if c.b_x:
        for kk in range(1, nkk):
            emit("v_xor_b32 %s, %d, %%[addrB]" % (BX(kk), kk * (32 if c.f64 else 16)))
    if not c.cont:
        for r in range(128, 128 + 16 * aw):
            emit("v_mov_b32 v%d, 0" % r)
    # stage 0 has landed for every wave -> stage 1 into buffer 1; first fragments
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    emit("s_cmp_lt_u32 %[rem], 2")
    emit("s_cbranch_scc1 7f")
    dma(1)
    emit("7:")
    reads(0, 0, 0)
    if c.diag is None:
        emit("1:")
        stage(0, "2f")
        stage(1, "2f")
        emit("s_branch 1b")
    elif c.diag == "sy":
        # diagonal tile of a SYRK-shaped launch (A and B are the same operand, only row block <= column block is
        # kept): wave (wr, wc) owns 16-row blocks wr + 2i and 16-column blocks wc + 2j, so sub-tile (i, j) lies on or
        # above the diagonal iff wr + 2i <= wc + 2j -- i <= j for three of the waves, i < j for wave (1, 0).  10 (6) of
        # the 16 MFMAs of a k-step remain; the strictly-lower sub-tiles keep their zeros.
        emit("s_cmp_eq_u32 %[wave], 2")
        emit("s_cbranch_scc1 20f")
        emit("1:")
        for b in (0, 1):
            stage(b, "2f", live=lambda kk: tuple((i, j) for i in range(4) for j in range(4) if i <= j))
        emit("s_branch 1b")
        emit("20:")
        for b in (0, 1):
            stage(b, "2f", live=lambda kk: tuple((i, j) for i in range(4) for j in range(4) if i < j))
        emit("s_branch 20b")
    elif c.diag in ("khi", "klod"):
        # plain stages until DSTAGES remain, then the diagonal block (a copy per starting buffer and wave column)
        def to_diag(label):
            def f():
                emit("s_cmp_eq_u32 %%[rem], %d" % DSTAGES)
                emit("s_cbranch_scc1 %s" % label)
            return f
        emit("s_cmp_eq_u32 %%[rem], %d" % DSTAGES)
        emit("s_cbranch_scc1 10f")
        emit("1:")
        stage(0, None, tail=to_diag("11f"))
        stage(1, None, tail=to_diag("10f"))
        emit("s_branch 1b")
        for b0, lab in ((0, 10), (1, 11)):
            emit("%d:" % lab)
            emit("s_bitcmp1_b32 %[wave], 0")
            emit("s_cbranch_scc1 %df" % (lab + 2))
            diag_section(b0, 0, "2f")
            emit("s_branch 2f")
            emit("%d:" % (lab + 2))
            diag_section(b0, 1, "2f")
            if b0 == 0:
                emit("s_branch 2f")
    else:
        # the diagonal block first (the k-range starts with it), then plain stages
        emit("s_bitcmp1_b32 %[wave], 0")
        emit("s_cbranch_scc1 12f")
        diag_section(0, 0, "2f")
        emit("s_branch 1f")
        emit("12:")
        diag_section(0, 1, "2f")
        emit("1:")
        stage(0, "2f")
        stage(1, "2f")
        emit("s_branch 1b")
    emit("2:")
    # the compiler's code reads the accumulators next: cover the last MFMA's write-back
    for _ in range(4):
        emit("s_nop 15")
    emit("s_mov_b32 m0, s92")
    return pro, list(L)


def main():
    out = ["// Generated by gen_engine_asm.py -- do not edit.", ""]
    for f64 in (True, False):
        for op, var, diag, cont, desc in (("nn", 0, None, False, False), ("nt", 0, None, False, False),
                                          ("tn", 0, None, False, False), ("tn", 1, None, False, False),
                                          ("tn", 2, None, False, False), ("nn", 0, "khi", False, False),
                                          ("nt", 0, "klo", False, False), ("tn", 1, "sy", False, False),
                                          ("tn", 2, "sy", False, False), ("nt", 0, None, True, False),
                                          ("nt", 0, "klo", True, False), ("nn", 0, None, False, True),
                                          ("nn", 0, "khid", False, False), ("nt", 0, None, False, True),
                                          ("nt", 0, "klod", False, False)):
            c = Cfg(f64, op, var, diag, cont, desc)
            for part, lines in zip(("PRO", "MAIN"), gen(c)):
                if (diag or cont) and part == "PRO":
                    continue   # same fetch as the plain variant (khid: as the plain descending one)
                out.append("#define %s \\" % c.name().replace("LOOP", part))
                for ln in lines:
                    out.append('  "%s\\n\\t" \\' % ln)
                out.append('  ""')
                out.append("")
    open(sys.argv[1] if len(sys.argv) > 1 else "engine_asm.inc", "w").write("\n".join(out))


if __name__ == "__main__":
    main()
