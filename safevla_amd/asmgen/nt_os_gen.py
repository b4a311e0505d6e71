"""Generator of the output-stationary NT GEMM kernels for gfx950:  C[M,N] (bf16) = A[M,K] . W[N,K]^T (+ bias[n]) (+ residual[m,n]),
M % 256 == 0, N % 256 == 0, K % 128 == 0, K >= 384.

Replaces gemm_nt8p_bf16_kernel of csrc/gemm.hip for the K > 512 linears of the fusion encoder without a dropout epilogue (reference:
nn.TransformerEncoder layers of architecture/models/allenact_transformer_models/allenact_dino_transformer.py:545-552,702-708 -- the input-gradient
GEMMs through linear1 (K = 2048) and in_proj (K = 1536), linear2's forward in eval mode).  With K > 512 nothing but the OUTPUT is stationary:

  * persistent, one workgroup of FOUR waves per CU (2 x 2, one per SIMD), each wave owns 128 x 128 of a 256 x 256 output tile: 16 MFMA 32x32
    accumulators = its 256 AGPRs; 0.5 LDS fragment reads per MFMA (ds_read_b128);
  * operands stream through a 128-KiB LDS ring of two K-tiles (64 k: every row of a unit is ONE full 128-byte line of A / W), in 4-KiB units of
    32 rows, XOR-swizzled on the LDS-DMA source address ((row >> 1) & 7: conflict-free ds_read_b128);
  * the K-tile is walked row block by row block: phase p = the 16 MFMAs (4 k-steps x 4 column blocks) of the wave's row block p.  The W fragments
    of the whole K-tile live in 64 VGPRs and are replaced in place during phase 3, each right behind its last MFMA; the A fragments of row block
    p + 1 are read at the head of phase p into the other of two 16-VGPR sets.  So the units of a K-tile are RELEASED one after the other (row
    block 0 first, W with row block 1), each is refilled with the data of two K-tiles ahead behind the next of the TWO barriers per K-tile, and
    every DMA piece has 1 ... 1.5 K-tiles (2.2 ... 3.3 k cycles) to land -- with whole K-tiles as the release unit the ring would hold only
    half as much data in flight;
  * the accumulators are all the registers there are, so the epilogue cannot run under the next tile's MFMAs -- but only its ARITHMETIC has to be
    exposed: accumulator -> [+ bias] [+ residual] -> bf16 pairs in 128 VGPRs (~ 1 k VALU instructions per tile).  The residual arrives before: row-major full-line loads
    under the second-to-last K-tile, straight into those 128 VGPRs, turned into the accumulator layout in place through the wave's 4-KiB LDS staging
    buffer under the last K-tile (8-byte accumulator-layout loads instead cost 7.5 k cycles per tile of address-unit time: measured).  The
    transposition of the result through the same buffer and the full-line non-temporal
    stores are DEFERRED into the MFMA gaps of the next tile's first two K-tiles (the operand DMA of the next tile runs on under all of it).
Every wait is counted by the generator's queue models and checked by amdasm.Emu (tests/test_asm_emulator_cpu.py).
"""
from .amdasm import M0, Prog, a, s, v
from .nt_as_gen import QModel

LDS_STG = 131072            # + wave * 4096
LDS_BIAS = 147456           # fp32 bias[N], N <= 4096
LDS_BYTES = 163840
PARB, WOFF, UNIT = 65536, 32768, 4096       # ring: [K-tile parity][A units 0..7 | W units 0..7][32 rows][128 B]
KARG = dict(A=0, lda=8, B=16, ldb=24, bias=32, res=40, ldr=48, C=56, ldc=64, M=72, N=76, K=80, ntn=84, ntiles=88, grid=92)
KARG_BYTES = 96

S_A, S_LDA, S_B, S_LDB, S_BIAS, S_RES, S_LDR, S_C = s(4, 2), s(6, 2), s(8, 2), s(10, 2), s(12, 2), s(14, 2), s(16, 2), s(18, 2)
S_LDC, S_M, S_N, S_K, S_NTN, S_NTILES, S_GRID = s(20, 2), s(22), s(23), s(24), s(25), s(26), s(27)
S_WID, S_WM, S_WN = s(28), s(29), s(30)
S_LDA2, S_LDB2, S_LDC2, S_LDR2 = s(31), s(32), s(33), s(34)
S_KT, S_DQ, S_DR = s(35), s(36), s(37)                 # K-tiles per output tile; grid / ntn, grid % ntn
S_TC, S_MTC, S_NTC, S_LOOP = s(38), s(39), s(40), s(41)        # tile being computed
S_TD, S_MTD, S_NTD, S_KLEFT = s(42), s(43), s(44), s(45)      # tile of the fetch cursor (K-tile T + 2), K-tiles left in it
S_X2, S_W2, S_X1, S_W1 = s(46, 2), s(48, 2), s(50, 2), s(52, 2)   # row 0 / k0 addresses of K-tiles T + 2 and T + 1
S_M0W, S_COL4 = s(54), s(55)
S_XRO = [s(56 + u) for u in range(8)]                  # (32 u + 8 w) * lda2: this wave's 8 rows of A unit u
S_WRO = [s(64 + u) for u in range(8)]
SRD_C, SRD_R, SRD_T = s(72, 4), s(76, 4), s(80, 4)
S_T = [s(84 + i) for i in range(12)]
S_RP = s(96, 2)
N_SGPR = 102


def ACC(nb, mb):
    return a((nb * 4 + mb) * 16, 16)


def WF(nb, ks):
    return v((nb * 4 + ks) * 4, 4)


def XF(st, ks):
    return v(64 + st * 16 + ks * 4, 4)


V_RDX = [[v(96 + par * 4 + ks) for ks in range(4)] for par in range(2)]
V_RDW = [[v(104 + par * 4 + ks) for ks in range(4)] for par in range(2)]
V_SX, V_SW, V_LANE, V_COFF, V_ROFF, _V_STRD0, V_BIASRD, V_BIASN = v(112), v(113), v(114), v(115), v(116), v(117), v(118), v(119)
V_STRD = [_V_STRD0, V_LANE]                           # row-major staging addresses, it even / odd (the lane id is dead after the prologue; set last)
V_STWA = [v(120 + gi) for gi in range(8)]              # staging addresses of the eight 8-byte pieces of a slab row (accumulator layout)


def OUT(slab, gi):
    """piece gi (4 bf16 of row lane & 31: columns 8 gi + 4 h ..) of slab (row block slab >> 1, 64-column half slab & 1): residual in, result out"""
    return v(128 + 16 * slab + 2 * gi, 2)


def OUTQ(slab, it):
    """the same 16 registers as the four row-major 16-byte pieces the read-back of the transposition returns"""
    return v(128 + 16 * slab + 4 * it, 4)


# temporaries of the exposed epilogue: the A fragment set 1 (dead behind the last MFMA of a K-tile; set 0 already holds the next K-tile's row block 0)
VS = [[v(80 + 4 * k + i) for i in range(4)] for k in range(2)]
BQ = [v(88, 4), v(92, 4)]
V_T = [v(248 + i) for i in range(8)]                   # prologue only
SRD_P = SRD_T                                          # C descriptor of the PREVIOUS tile (deferred stores); the bias-table descriptor in the prologue


class NtOsGen:
    def __init__(self, name="svla_nt_os_p", bias=False, res=False, dbg=""):
        self.name, self.bias, self.res = name, bias, res
        self.dbg = set(dbg.split(",")) if dbg else set()
        self.p = Prog(name)
        self.vm = QModel(63)
        self.lg = QModel(15)
        self.uid = 0

    # ------------------------------------------------------------------ helpers
    def lab(self, base):
        self.uid += 1
        return f"L_{base}{self.uid}"

    def wait_for(self, vm_tags=(), lg_tags=()):
        nv = self.vm.need(set(vm_tags)) if vm_tags else None
        nl = self.lg.need(set(lg_tags)) if lg_tags else None
        if nv is None and nl is None:
            return
        self.p.s_waitcnt(vmcnt=nv, lgkmcnt=nl)
        if nv is not None:
            self.vm.wait(nv)
        if nl is not None:
            self.lg.wait(nl)

    def lg_room(self):
        if len(self.lg.q) >= 15:
            self.p.s_waitcnt(lgkmcnt=11)
            self.lg.wait(11)

    def ds_read(self, d, addr, off, tag):
        self.lg_room()
        self.p.ds_read(d, addr, off)
        self.lg.issue(tag)

    def ds_write(self, addr, src, off=0):
        self.lg_room()
        self.p.ds_write(addr, src, off)
        self.uid += 1
        self.lg.issue(f"dsw#{self.uid}")

    def divmod(self, q, r, num, den):
        """q, r = num / den, num % den by repeated subtraction (small quotients; runs once per kernel)"""
        p = self.p
        top, done = self.lab("DIV"), self.lab("DIVD")
        p.s_mov_b32(q, 0)
        p.s_mov_b32(r, num)
        p.label(top)
        p.s_cmp("lt_u32", r, den)
        p.s_cbranch_scc1(done)
        p.s_sub_u32(r, r, den)
        p.s_add_u32(q, q, 1)
        p.s_branch(top)
        p.label(done)

    def tile_ptrs(self, xdst, wdst, mt, nt):
        """addresses of (row 0, k = 0) of the A row tile mt and of the W row tile nt"""
        p = self.p
        for dst, base, idx, ld2 in ((xdst, S_A, mt, S_LDA2), (wdst, S_B, nt, S_LDB2)):
            p.s_lshl_b32(S_T[0], idx, 8)
            p.s_mul_hi_u32(S_T[1], S_T[0], ld2)
            p.s_mul_i32(S_T[0], S_T[0], ld2)
            p.s_add_u32(dst.sub(0), base.sub(0), S_T[0])
            p.s_addc_u32(dst.sub(1), base.sub(1), S_T[1])

    def cursor_next(self):
        """K-tile T + 1 <- T + 2; T + 2 <- the K-tile after it (the next 64 k of its output tile, or the first of this workgroup's next tile;
        past the last tile the cursor wraps onto that tile again: fetched, never computed)"""
        p = self.p
        adv, stay, done = self.lab("ADV"), self.lab("STAY"), self.lab("CUR")
        for i in range(2):
            p.s_mov_b32(S_X1.sub(i), S_X2.sub(i))
            p.s_mov_b32(S_W1.sub(i), S_W2.sub(i))
        p.s_sub_u32(S_KLEFT, S_KLEFT, 1)
        p.s_cmp("eq_u32", S_KLEFT, 0)
        p.s_cbranch_scc1(adv)
        for ptr in (S_X2, S_W2):
            p.s_add_u32(ptr.sub(0), ptr.sub(0), 128)
            p.s_addc_u32(ptr.sub(1), ptr.sub(1), 0)
        p.s_branch(done)
        p.label(adv)
        p.s_add_u32(S_T[0], S_TD, S_GRID)
        p.s_cmp("lt_u32", S_T[0], S_NTILES)
        p.s_cbranch_scc0(stay)
        p.s_mov_b32(S_TD, S_T[0])
        p.s_add_u32(S_MTD, S_MTD, S_DQ)
        p.s_add_u32(S_NTD, S_NTD, S_DR)
        p.s_cmp("ge_u32", S_NTD, S_NTN)
        p.s_cselect_b32(S_T[0], S_NTN, 0)
        p.s_cselect_b32(S_T[1], 1, 0)
        p.s_sub_u32(S_NTD, S_NTD, S_T[0])
        p.s_add_u32(S_MTD, S_MTD, S_T[1])
        p.label(stay)
        self.tile_ptrs(S_X2, S_W2, S_MTD, S_NTD)
        p.s_mov_b32(S_KLEFT, S_KT)
        p.label(done)

    def dma(self, which, cur, par, u):
        """scheduler group: this wave's LDS-DMA instruction (8 rows) of unit u of operand 'x' / 'w' of the K-tile at cursor cur, into ring parity par"""
        p = self.p
        ro = (S_XRO if which == "x" else S_WRO)[u]

        def g():
            p.s_add_u32(S_RP.sub(0), cur.sub(0), ro)
            p.s_addc_u32(S_RP.sub(1), cur.sub(1), 0)
            p.s_add_u32(M0, S_M0W, par * PARB + (WOFF if which == "w" else 0) + u * UNIT)
            p.s_nop(0)
            if "nodma" not in self.dbg:
                p.global_load_lds_x4(V_SX if which == "x" else V_SW, S_RP)      # (nt: -5 ... -8 %, the two n-tiles of a row tile share A in the L2; sc1: +-0)
                self.vm.issue(f"{which}{par}_{u}")
        return g

    def xread(self, st, par, mb, ks):
        self.ds_read(XF(st, ks), V_RDX[par][ks], mb * UNIT, f"xf{st}_{ks}")

    def wread(self, par, nb, ks):
        self.ds_read(WF(nb, ks), V_RDW[par][ks], nb * UNIT, f"wf{nb}_{ks}")

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        T = V_T
        p.s_load(s(4, 16), s(0, 2), 0)
        p.s_load(s(20, 8), s(0, 2), 64)
        p.v_and_b32(V_LANE, 63, v(0))
        p.v_lshrrev_b32(T[1], 6, v(0))
        p.v_readfirstlane_b32(S_WID, T[1])
        p.s_waitcnt(lgkmcnt=0)
        p.s_lshr_b32(S_WM, S_WID, 1)
        p.s_and_b32(S_WN, S_WID, 1)
        p.s_lshl_b32(S_LDA2, S_LDA.sub(0), 1)
        p.s_lshl_b32(S_LDB2, S_LDB.sub(0), 1)
        p.s_lshl_b32(S_LDC2, S_LDC.sub(0), 1)
        p.s_lshl_b32(S_LDR2, S_LDR.sub(0), 1)
        p.s_lshr_b32(S_KT, S_K, 6)
        p.s_lshl_b32(S_M0W, S_WID, 10)
        # ---- workgroup -> first tile: XCD-contiguous virtual index (workgroup b runs on XCD b % 8): the n-tiles of a row tile share an L2
        p.s_mov_b32(S_T[0], s(2))
        p.s_lshr_b32(S_T[1], S_GRID, 3)             # q = grid / 8
        p.s_and_b32(S_T[2], S_GRID, 7)              # r = grid % 8
        p.s_and_b32(S_T[3], S_T[0], 7)              # xcd
        p.s_lshr_b32(S_T[4], S_T[0], 3)             # slot
        p.s_add_u32(S_T[5], S_T[1], 1)
        lo, xd = self.lab("XLO"), self.lab("XD")
        p.s_cmp("lt_u32", S_T[3], S_T[2])
        p.s_cbranch_scc1(lo)
        p.s_mul_i32(S_T[6], S_T[2], S_T[5])
        p.s_sub_u32(S_T[7], S_T[3], S_T[2])
        p.s_mul_i32(S_T[7], S_T[7], S_T[1])
        p.s_add_u32(S_T[6], S_T[6], S_T[7])
        p.s_branch(xd)
        p.label(lo)
        p.s_mul_i32(S_T[6], S_T[3], S_T[5])
        p.label(xd)
        p.s_add_u32(S_TC, S_T[6], S_T[4])           # vid
        self.divmod(S_MTC, S_NTC, S_TC, S_NTN)
        self.divmod(S_DQ, S_DR, S_GRID, S_NTN)
        p.s_mov_b32(S_TD, S_TC)
        p.s_mov_b32(S_MTD, S_MTC)
        p.s_mov_b32(S_NTD, S_NTC)
        # ---- row offsets of this wave's share (rows 8 w .. 8 w + 7) of every unit
        p.s_lshl_b32(S_T[0], S_WID, 3)
        for u in range(8):
            p.s_add_u32(S_T[1], S_T[0], 32 * u)
            p.s_mul_i32(S_XRO[u], S_T[1], S_LDA2)
            p.s_mul_i32(S_WRO[u], S_T[1], S_LDB2)
        # ---- lane constants
        # LDS-DMA source offsets: lane i lands at (row i >> 3, physical 16-byte chunk i & 7) of the 8 rows; physical = logical ^ ((row >> 1) & 7)
        # with row = 8 w + (i >> 3) inside the unit
        p.v_lshrrev_b32(T[2], 3, V_LANE)            # i >> 3
        p.v_and_b32(T[3], 7, V_LANE)                # i & 7
        p.v_lshrrev_b32(T[4], 4, V_LANE)            # i >> 4
        p.s_lshl_b32(S_T[1], S_WID, 2)
        p.v_add_u32(T[4], S_T[1], T[4])
        p.v_and_b32(T[4], 7, T[4])                  # (4 w + (i >> 4)) & 7
        p.v_xor_b32(T[3], T[3], T[4])               # logical chunk
        p.v_lshlrev_b32(T[3], 4, T[3])
        p.v_mul_lo_u32(T[5], T[2], S_LDA2)
        p.v_add_u32(V_SX, T[5], T[3])
        p.v_mul_lo_u32(T[5], T[2], S_LDB2)
        p.v_add_u32(V_SW, T[5], T[3])
        # fragment reads: row r = lane & 31 of the unit, k-step ks: logical chunk 2 ks + h at physical chunk ^ ((r >> 1) & 7)
        p.v_and_b32(T[2], 31, V_LANE)               # r
        p.v_lshrrev_b32(T[3], 5, V_LANE)            # h
        p.v_lshrrev_b32(T[4], 1, T[2])
        p.v_and_b32(T[4], 7, T[4])
        p.v_lshlrev_b32(T[5], 7, T[2])              # r * 128
        p.s_lshl_b32(S_T[1], S_WM, 14)
        p.s_lshl_b32(S_T[2], S_WN, 14)
        p.s_add_u32(S_T[2], S_T[2], WOFF)
        for ks in range(4):
            p.v_add_u32(T[6], 2 * ks, T[3])
            p.v_xor_b32(T[6], T[6], T[4])
            p.v_lshl_add_u32(T[6], T[6], 4, T[5])
            for par in range(2):
                p.v_add_u32(V_RDX[par][ks], S_T[1], T[6])
                p.v_add_u32(V_RDW[par][ks], S_T[2], T[6])
                if par:
                    p.v_add_u32(V_RDX[par][ks], PARB, V_RDX[par][ks])
                    p.v_add_u32(V_RDW[par][ks], PARB, V_RDW[par][ks])
        # epilogue: row-major side (lane -> row lane >> 3 (+ 8 it), 16-byte chunk lane & 7), accumulator side (lane -> row c = lane & 31, 8 bytes
        # at column 4 h of piece gi); staging [32 rows][128 B], 16-byte chunk q of row r at chunk q ^ (r & 7)
        p.v_lshrrev_b32(T[6], 3, V_LANE)
        p.v_and_b32(T[7], 7, V_LANE)
        p.v_mul_lo_u32(T[5], T[6], S_LDC2)
        p.v_lshl_add_u32(V_COFF, T[7], 4, T[5])
        p.v_mul_lo_u32(T[5], T[6], S_LDR2)
        p.v_lshl_add_u32(V_ROFF, T[7], 4, T[5])
        p.s_lshl_b32(S_T[3], S_WID, 12)
        p.s_add_u32(S_T[3], S_T[3], LDS_STG)
        # staging swizzle: the 16-byte chunk q of row r sits at chunk q ^ ((r >> 1) & 7), and in rows 16-31 the two 8-byte halves of a chunk are
        # swapped: the 32 lanes of an accumulator-layout access (8 bytes each, rows 0-31, one chunk) then cover 32 different 8-byte slots of the 256
        # bytes the LDS serves per clock (with q ^ (r & 7) alone, rows r, r + 8, r + 16, r + 24 met in one slot: 13 % of the LDS cycles were conflicts)
        p.v_lshrrev_b32(T[4], 4, V_LANE)            # row-major side: row R = (lane >> 3) + 8 it: (R >> 1) & 7 = ((lane >> 4) + 4 it) & 7
        for par in range(2):
            p.v_add_u32(T[5], 4 * par, T[4])
            p.v_xor_b32(T[5], T[7], T[5])
            p.v_lshlrev_b32(T[1], 7, T[6])
            p.v_lshl_add_u32(T[1], T[5], 4, T[1])
            p.v_add_u32(V_STRD[par], S_T[3], T[1])
        p.v_lshrrev_b32(T[4], 1, T[2])
        p.v_and_b32(T[4], 7, T[4])                  # (c >> 1) & 7
        p.v_lshrrev_b32(T[1], 4, T[2])              # c >> 4
        p.v_xor_b32(T[1], T[1], T[3])               # half: h ^ (c >> 4)
        p.v_lshlrev_b32(T[5], 7, T[2])              # c * 128
        p.v_lshl_add_u32(T[5], T[1], 3, T[5])
        p.v_add_u32(T[5], S_T[3], T[5])
        for gi in range(8):
            p.v_xor_b32(T[6], gi, T[4])
            p.v_lshl_add_u32(V_STWA[gi], T[6], 4, T[5])
        p.v_lshlrev_b32(T[6], 4, T[3])
        p.v_add_u32(V_BIASRD, LDS_BIAS, T[6])       # + 16 h
        for srd in (SRD_C, SRD_R):
            p.s_mov_b32(srd.sub(2), 0xffffffff)
            p.s_mov_b32(srd.sub(3), 0x00020000)
        if self.bias:
            self.bias_table()
        for i in range(4):
            p.s_mov_b32(SRD_P.sub(i), [0, 0, 0, 0x00020000][i])      # num_records = 0: the deferred stores of "the tile before the first" are dropped
        # ---- ring fill: K-tile 0 whole, K-tile 1 without the A row blocks 3 (they follow the first barrier of K-tile 0)
        self.tile_ptrs(S_X2, S_W2, S_MTD, S_NTD)
        p.s_mov_b32(S_KLEFT, S_KT)
        for u in range(8):
            self.dma("x", S_X2, 0, u)()
            self.dma("w", S_W2, 0, u)()
        self.cursor_next()
        for u in range(8):
            if u & 3 != 3:
                self.dma("x", S_X2, 1, u)()
            self.dma("w", S_W2, 1, u)()
        self.cursor_next()
        self.tile_setup()
        p.s_lshr_b32(S_LOOP, S_KT, 1)
        p.s_sub_u32(S_LOOP, S_LOOP, 2)
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        self.vm.wait(0)
        self.lg.wait(0)
        p.s_barrier()
        for ks in range(4):
            self.xread(0, 0, 0, ks)
        for nb in range(4):
            for ks in range(4):
                self.wread(0, nb, ks)
        p.s_waitcnt(lgkmcnt=0)
        self.lg.wait(0)

    def bias_table(self):
        p = self.p
        T = V_T
        p.s_mov_b32(SRD_T.sub(0), S_BIAS.sub(0))
        p.s_and_b32(SRD_T.sub(1), S_BIAS.sub(1), 0xffff)
        p.s_mov_b32(SRD_T.sub(3), 0x00020000)
        p.v_lshlrev_b32(T[0], 2, v(0))
        p.v_add_u32(T[1], LDS_BIAS, T[0])
        p.s_mov_b32(S_T[3], 0)
        p.s_lshl_b32(S_T[4], S_N, 2)
        p.s_mov_b32(SRD_T.sub(2), S_T[4])
        top = self.lab("BIAS")
        p.label(top)
        p.buffer_load(T[2], T[0], SRD_T, S_T[3])
        p.s_waitcnt(vmcnt=0)
        p.ds_write(T[1], T[2])
        p.v_add_u32(T[1], 1024, T[1])
        p.s_add_u32(S_T[3], S_T[3], 1024)
        p.s_cmp("lt_u32", S_T[3], S_T[4])
        p.s_cbranch_scc1(top)

    def tile_setup(self):
        """descriptors of the tile being computed: C / residual at (row mt*256 + wm*128, column nt*256 + wn*128)"""
        p = self.p
        p.s_lshl_b32(S_T[0], S_MTC, 8)
        p.s_lshl_b32(S_T[1], S_WM, 7)
        p.s_add_u32(S_T[0], S_T[0], S_T[1])         # row
        p.s_lshl_b32(S_T[2], S_NTC, 8)
        p.s_lshl_b32(S_T[1], S_WN, 7)
        p.s_add_u32(S_T[2], S_T[2], S_T[1])         # column
        p.s_lshl_b32(S_COL4, S_T[2], 2)
        p.s_lshl_b32(S_T[2], S_T[2], 1)
        for srd, base, ld2 in ((SRD_C, S_C, S_LDC2),) + (((SRD_R, S_RES, S_LDR2),) if self.res else ()):
            p.s_mul_hi_u32(S_T[4], S_T[0], ld2)
            p.s_mul_i32(S_T[3], S_T[0], ld2)
            p.s_add_u32(S_T[3], S_T[3], S_T[2])
            p.s_addc_u32(S_T[4], S_T[4], 0)
            p.s_add_u32(srd.sub(0), base.sub(0), S_T[3])
            p.s_addc_u32(S_T[4], base.sub(1), S_T[4])
            p.s_and_b32(srd.sub(1), S_T[4], 0xffff)
        if self.bias:
            p.v_add_u32(V_BIASN, S_COL4, V_BIASRD)

    # ------------------------------------------------------------------ one K-tile (64 MFMAs per wave)
    def res_loads(self, slab):
        """scheduler groups: the four row-major (full-line) loads of a residual slab into the registers its result will leave from"""
        p = self.p
        mb, j = slab >> 1, slab & 1
        groups = []
        for it in range(4):
            def g(it=it):
                p.s_mul_i32(S_T[5], S_LDR2, mb * 32 + 8 * it)
                p.buffer_load(OUTQ(slab, it), V_ROFF, SRD_R, S_T[5], j * 128)
                self.vm.issue(f"rs{slab}_{it}")
            groups.append(g)
        return groups

    def res_transpose(self, slab):
        """scheduler groups: residual slab, row-major registers -> staging -> the same registers in the accumulator layout (LDS operations of a wave execute
        in order: the next slab's writes follow this slab's reads without a wait)"""
        def w():
            for it in range(4):
                self.wait_for(vm_tags=[f"rs{slab}_{it}"])
                if it >= 2:         # rows 16-31 keep the halves of a chunk swapped
                    self.p.v_swap_b32(OUTQ(slab, it).sub(0), OUTQ(slab, it).sub(2))
                    self.p.v_swap_b32(OUTQ(slab, it).sub(1), OUTQ(slab, it).sub(3))
                self.ds_write(V_STRD[it & 1], OUTQ(slab, it), it * 1024)

        def r(g0):
            def f():
                for gi in range(g0, g0 + 4):
                    self.ds_read(OUT(slab, gi), V_STWA[gi], 0, f"rp{slab}")
            return f
        return [w, r(0), r(4)]

    def drain_groups(self, slab):
        """scheduler groups: the deferred half of the epilogue for one slab of the PREVIOUS tile -- bf16 pieces -> staging (accumulator layout in, row-major
        out, back into the same registers) -> four full-line stores"""
        p = self.p
        mb, j = slab >> 1, slab & 1
        groups = []
        for g0 in range(0, 8, 2):
            def w(g0=g0):
                for gi in (g0, g0 + 1):
                    self.ds_write(V_STWA[gi], OUT(slab, gi))
            groups.append(w)
        for i0 in range(0, 4, 2):
            def r(i0=i0):
                for it in (i0, i0 + 1):
                    self.ds_read(OUTQ(slab, it), V_STRD[it & 1], it * 1024, f"rb{slab}_{it}")
            groups.append(r)
        for it in range(4):
            def st(it=it):
                self.wait_for(lg_tags=[f"rb{slab}_{it}"])
                if it >= 2:
                    p.v_swap_b32(OUTQ(slab, it).sub(0), OUTQ(slab, it).sub(2))
                    p.v_swap_b32(OUTQ(slab, it).sub(1), OUTQ(slab, it).sub(3))
                p.s_mul_i32(S_T[5], S_LDC2, mb * 32 + 8 * it)
                if "nostore" not in self.dbg:
                    p.buffer_store(OUTQ(slab, it), V_COFF, SRD_P, S_T[5], j * 128, nt=True)
                    self.uid += 1
                    self.vm.issue(f"st#{self.uid}")
            groups.append(st)
        return groups

    def body(self, kind, par):
        """K-tile T of parity par.  kind: 'first' (of an output tile: the accumulators start from 0; drains slabs 0-3 of the previous tile), 'second'
        (drains slabs 4-7), 'mid', 'prelast' / 'last' (fetch the residual of slabs 0-3 / 4-7)"""
        p = self.p
        NG = 64
        fixed = [[] for _ in range(NG)]
        # A fragments of the next row block at the head of every phase (other register set); the next K-tile's row block 0 in phase 3
        for ph in range(4):
            for ks in range(4):
                if ph < 3:
                    fixed[16 * ph + ks].append(lambda ph=ph, ks=ks: self.xread((ph + 1) & 1, par, ph + 1, ks))
                else:
                    fixed[48 + ks].append(lambda ks=ks: self.xread(0, par ^ 1, 0, ks))
        # W fragments of the next K-tile, each right behind its last MFMA of this one
        for ks in range(4):
            for nb in range(4):
                fixed[48 + 4 * ks + nb].append(lambda nb=nb, ks=ks: self.wread(par ^ 1, nb, ks))
        # DMA: behind the first barrier A(T + 1, row block 3) and A(T + 2, row block 0); behind the second W(T + 2) and A(T + 2, row blocks 1, 2)
        s0 = [self.dma("x", S_X1, par ^ 1, wm * 4 + 3) for wm in range(2)] + [self.dma("x", S_X2, par, wm * 4) for wm in range(2)]
        s1 = [self.dma("w", S_W2, par, u) for u in range(8)] + [self.dma("x", S_X2, par, wm * 4 + mb) for mb in (1, 2) for wm in range(2)]
        streams = [(s0, 1, 30), (s1, 33, 62)]
        epi = "noepi" not in self.dbg
        # the residual of the tile being computed goes into the output registers as soon as the previous tile's pieces have left them: slabs 0-5 under
        # the second K-tile, 6-7 under the second-to-last (at least one pair of the loop lies in between: K >= 384) -- an HBM round trip is longer than a K-tile
        res = self.res and epi
        if kind == "first" and epi:
            streams.append(([g for slab in range(4) for g in self.drain_groups(slab)], 2, 61))
        if kind == "second" and epi:
            d = {slab: self.drain_groups(slab) for slab in range(4, 8)}
            ld = {slab: (self.res_loads(slab) if res else []) for slab in range(8)}
            streams.append((d[4] + d[5] + ld[4] + d[6] + ld[5] + d[7], 2, 61))
            if res:
                streams.append(([g for slab in range(4) for g in ld[slab]], 2, 40))
        if kind == "prelast" and res:
            streams.append(([g for slab in (6, 7) for g in self.res_loads(slab)], 2, 12))
        if kind == "last" and self.res and epi:
            streams.append(([g for slab in range(8) for g in self.res_transpose(slab)], 20, 61))
        pos = [0] * len(streams)
        for g in range(NG):
            ph, ks, nb = g >> 4, (g >> 2) & 3, g & 3
            if g == 0:
                # all reads of A(T - 1, 3) and A(T, 0) have retired (the latter are this phase's operands); A(T, 1), A(T, 2) have landed
                self.barrier([f"x{par}_{wm * 4 + mb}" for mb in (1, 2) for wm in range(2)], [f"xf0_{k}" for k in range(4)])
            if g == 32:
                # reads of A(T, 1), A(T, 2) retired, W(T) consumed; A(T, 3), A(T + 1, 0), W(T + 1) have landed
                self.barrier([f"x{par}_{wm * 4 + 3}" for wm in range(2)] + [f"x{par ^ 1}_{wm * 4}" for wm in range(2)] +
                             [f"w{par ^ 1}_{u}" for u in range(8)], [f"xf0_{k}" for k in range(4)])
            self.wait_for(lg_tags=[f"xf{ph & 1}_{ks}", f"wf{nb}_{ks}"])
            c = 0 if (kind == "first" and ks == 0) else ACC(nb, ph)
            if "nomfma" not in self.dbg:
                p.v_mfma_f32_32x32x16_bf16(ACC(nb, ph), WF(nb, ks), XF(ph & 1, ks), c)
            for th in fixed[g]:
                th()
            for si, (sm, g0, g1) in enumerate(streams):
                due = len(sm) if g >= g1 else (0 if g < g0 else (len(sm) * (g - g0 + 1) + (g1 - g0)) // (g1 - g0 + 1))
                while pos[si] < due:
                    sm[pos[si]]()
                    pos[si] += 1
        self.cursor_next()

    def barrier(self, vm_tags, lg_tags):
        p = self.p
        nv = self.vm.need(set(vm_tags))
        nl = self.lg.need(set(lg_tags))
        if nv is not None or nl is not None:
            p.s_waitcnt(vmcnt=nv, lgkmcnt=nl)
            if nv is not None:
                self.vm.wait(nv)
            if nl is not None:
                self.lg.wait(nl)
        p.s_barrier()

    # ------------------------------------------------------------------ exposed half of the epilogue: accumulators -> bf16 pieces
    def pack(self):
        """Two quads in flight (two temporary sets): the result of a DOT instruction may not be read by another kind of VALU instruction for 3 wait
        states (the hardware does not interlock; amdasm.Emu models it), so the conversion of quad q follows the arithmetic of quad q + 1."""
        p = self.p
        if "noepi" in self.dbg:
            return
        p.s_nop(7)          # the last MFMAs are still reading A fragment set 1: these temporaries live there
        p.s_nop(7)
        if self.res:
            p.s_mov_b32(S_T[6], 0x00003f80)      # bf16 (1, 0): picks the low half of a pair
            p.s_mov_b32(S_T[7], 0x3f800000)
        quads = [(slab, gi) for slab in range(8) for gi in range(8)]

        def bias_read(q):
            slab, gi = quads[q]
            nb, rg = 2 * (slab & 1) + (gi >> 2), gi & 3
            self.ds_read(BQ[q & 1], V_BIASN, (nb * 32 + rg * 8) * 4, f"bq{q & 1}")

        def arith(q):
            slab, gi = quads[q]
            mb, j = slab >> 1, slab & 1
            nb, rg = 2 * j + (gi >> 2), gi & 3
            acc, vs = ACC(nb, mb), VS[q & 1]
            if self.bias and q + 1 < len(quads):
                bias_read(q + 1)
            for e in range(4):
                p.v_accvgpr_read_b32(vs[e], acc.sub(4 * rg + e))
            if self.bias:
                self.wait_for(lg_tags=[f"bq{q & 1}"])
                for e in range(4):
                    p.v_add_f32(vs[e], BQ[q & 1].sub(e), vs[e])
            if self.res:
                if gi == 0:
                    self.wait_for(lg_tags=[f"rp{slab}"])
                for e in range(4):      # += the bf16 half e & 1 of the pair: a dot product with (1, 0) / (0, 1)
                    p.v_dot2c_f32_bf16(vs[e], S_T[6 + (e & 1)], OUT(slab, gi).sub(e >> 1))

        def convert(q):
            slab, gi = quads[q]
            vs = VS[q & 1]
            p.v_cvt_pk_bf16_f32(OUT(slab, gi).sub(0), vs[0], vs[1])
            p.v_cvt_pk_bf16_f32(OUT(slab, gi).sub(1), vs[2], vs[3])
        if self.bias:
            bias_read(0)
        for q in range(len(quads)):
            arith(q)
            if q:
                convert(q - 1)
        if self.res:
            p.s_nop(2)
        convert(len(quads) - 1)

    # ------------------------------------------------------------------ whole kernel
    def scratch(self, *bodies):
        """advance the queue models over a code sequence without emitting it (what precedes a body that is entered from later code at run time)"""
        real, self.p = self.p, Prog("scratch")
        for b in bodies:
            b()
        self.p = real

    def build(self):
        p = self.p
        self.prologue()
        # the first body is entered from the exposed epilogue at run time (and once from the prologue, where every operation it could wait for has
        # retired).  Wherever two paths lead to a body its waits are counted for the path that leaves the FEWEST operations in the queues: on the
        # others the same counts retire at least as much.
        self.scratch(lambda: self.body("prelast", 0), lambda: self.body("last", 1), self.pack)
        p.label("L_TILE")
        self.body("first", 0)
        self.body("second", 1)
        self.scratch(lambda: self.body("mid", 0), lambda: self.body("mid", 1))
        p.label("L_PAIR")
        self.body("mid", 0)
        self.body("mid", 1)
        p.s_sub_u32(S_LOOP, S_LOOP, 1)
        p.s_cmp("lg_u32", S_LOOP, 0)
        p.s_cbranch_scc1("L_PAIR")
        self.body("prelast", 0)
        self.body("last", 1)
        self.pack()
        # next tile of this workgroup; the tile just finished becomes "the previous tile" of the deferred stores
        for i in range(4):
            p.s_mov_b32(SRD_P.sub(i), SRD_C.sub(i))
        p.s_add_u32(S_TC, S_TC, S_GRID)
        p.s_cmp("ge_u32", S_TC, S_NTILES)
        p.s_cbranch_scc1("L_EXIT")
        p.s_add_u32(S_MTC, S_MTC, S_DQ)
        p.s_add_u32(S_NTC, S_NTC, S_DR)
        p.s_cmp("ge_u32", S_NTC, S_NTN)
        p.s_cselect_b32(S_T[0], S_NTN, 0)
        p.s_cselect_b32(S_T[1], 1, 0)
        p.s_sub_u32(S_NTC, S_NTC, S_T[0])
        p.s_add_u32(S_MTC, S_MTC, S_T[1])
        self.tile_setup()
        p.s_lshr_b32(S_LOOP, S_KT, 1)
        p.s_sub_u32(S_LOOP, S_LOOP, 2)
        p.s_branch("L_TILE")
        p.label("L_EXIT")
        # the last tile's deferred half, nothing to hide it under
        p.s_waitcnt(lgkmcnt=0)
        self.lg.wait(0)
        if "noepi" not in self.dbg:
            for slab in range(8):
                for g in self.drain_groups(slab):
                    g()
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        p.s_endpgm()
        return self

    def asm_text(self):
        from . import nt_as_gen as G
        g = G.NtAsGen(name=self.name)
        g.p = self.p
        t = g.asm_text()
        t = t.replace(f".amdhsa_kernarg_size {G.KARG_BYTES}", f".amdhsa_kernarg_size {KARG_BYTES}").replace(
            f".kernarg_segment_size: {G.KARG_BYTES}", f".kernarg_segment_size: {KARG_BYTES}").replace(
            f".size: {G.KARG_BYTES}, .offset: 0", f".size: {KARG_BYTES}, .offset: 0")
        return t.replace("nt_as_gen.py", "nt_os_gen.py")


# flavour -> generator options (the C dispatcher nt_os_try of csrc/gemm.hip picks by name)
FLAVOURS = {"p": dict(), "b": dict(bias=True), "r": dict(res=True), "br": dict(bias=True, res=True)}


import os as _os
if _os.environ.get("SVLA_ASM_DEBUG_VARIANTS"):      # timing-only builds (tools/var_nt_os.py): wrong results
    for _d in ("noepi", "nodma", "nomfma", "nostore", "noepi,nodma", "noepi,nomfma"):
        FLAVOURS["r_" + _d.replace(",", "_")] = dict(res=True, dbg=_d)
        FLAVOURS["p_" + _d.replace(",", "_")] = dict(dbg=_d)


def generate(flavour="p"):
    o = dict(FLAVOURS[flavour])
    return NtOsGen(name=f"svla_nt_os_{flavour}", dbg=o.pop("dbg", ""), **o).build()
