"""Generator of the A-stationary NT GEMM kernels for gfx950 (C[M,N] = epi(A[M,K] . W[N,K]^T), K = 512 or 384, M % 256 == 0, N % 128 == 0).

Replaces, for the K = 512 linears of the fusion encoder (reference: nn.TransformerEncoder layers of
architecture/models/allenact_transformer_models/allenact_dino_transformer.py:545-552,702-708 -- in_proj, out_proj, linear1, and the
input-gradient GEMM through linear2), the 256x256-tile kernel of csrc/gemm.hip.  Structure:

  * one workgroup of FOUR waves per CU (one wave per SIMD, 256 arch VGPRs + 256 AGPRs each), persistent over 256-row panels of A;
  * wave w keeps ITS 64 rows x 512 k of A in the 256 AGPRs for a whole sweep over N (MFMA B operands read straight from the accumulator
    file): A is fetched from HBM exactly once and never touches LDS;
  * W streams L2 -> LDS by LDS-DMA, one 64-column x 512-k tile (64 KiB, rows XOR-swizzled on the DMA source address) per "n-step",
    two buffers, ONE workgroup barrier per n-step (4096 MFMA cycles); the tile sequence 0, 1, .., N/64 - 1, 0, .. is shared by the
    workgroup and never restarts;
  * per n-step a wave issues 128 v_mfma_f32_32x32x16_bf16 (2 n-blocks x 2 m-blocks x 32 k-steps) against 64 ds_read_b128: 0.5 LDS
    fragment reads per MFMA;
  * TWO accumulator sets (2 x 64 VGPRs): while n-step j accumulates into one, the epilogue of n-step j-1 (bias is the accumulator's
    initial value, activation / dropout / mask / residual, bf16 rounding, transposition through a wave-private 4-KiB LDS buffer, eight
    full-line stores) is interleaved, instruction by instruction, into the gaps between the MFMAs of step j -- the C stores leave the CU
    as a continuous trickle instead of a burst at the end of a 256x256 tile;
  * PHASES: a wave's sweep over N may start at any tile of the shared sequence (the n-tiles of a panel are visited in a rotated order),
    so wave w of workgroup g switches panels phi(g, w) = w * NS/4 + (g & cmask) steps after wave 0 of workgroup 0 does (it idles through
    phi "dummy" steps at the start of the launch and (3 - w) * NS/4 at the end, serving the DMA and the barriers).  The next panel's A
    fragments are fetched in the panel's last step, fragment by fragment as their last MFMA has been issued; with every wave of every
    CU switching in the same step that was a chip-wide burst of 256 KiB per CU (measured: the last step took 22.8 k cycles instead of 5 k);
    with phases the chip fetches A at a constant rate;
  * every s_waitcnt is counted by this generator from a model of the in-order VM / LGKM queues; amdasm.Emu checks the result.

Round 5: (a) MID-M launches -- grid = (panel slots) x (n-ranges): workgroup_id_y = y sweeps columns [y nr, (y + 1) nr) of its panels (kernargs nr / flags; the
prologue moves B / bias / C / sign-bit / dropout offsets to the range's first column, S_N becomes the range's width, S_NFULL keeps N for the global indexing), flags
bit 0 switches the phases off (one panel per workgroup has nothing to de-phase); (b) K = 384 (24 k-steps: the gap positions of the schedule scale, a W row is 768
bytes = 48 DMA lanes under an exec mask, LDS pitch unchanged); (c) the GELU flavour (asmgen/gelu_poly.py as packed-fp32 Horner chains in the MFMA gaps).

The instruction stream is emitted by a list scheduler: each MFMA is followed by up to CAP "filler" groups taken from fixed slots
(W fragment reads) and from ordered streams (LDS-DMA of the next W tile, epilogue of the previous step).  A group is emitted
whole (carry chains through SCC, M0 write + DMA).
"""
import os as _os

from . import gelu_poly
from .amdasm import EXEC, M0, Prog, a, s, v

KS_DEFAULT = 32            # k-steps of 16 (K = 512); the ViT-S flavours run K = 384 (24 k-steps: 192 of the 256 AGPRs hold A)
LDS_W = (0, 65536)
LDS_STG = 131072           # + wave * 4096
LDS_BIAS = 147456          # fp32 bias[N], N <= 4096 (the bias table doubles as the accumulator initialiser)
LDS_BYTES = 163840

# ---- kernel arguments (byte offsets in the kernarg segment)
# nr: columns of the n-range one workgroup sweeps (= N for the row-streaming launches; N / nsplit for the mid-M launches whose grid is (panel slots,
# nsplit): workgroup_id_y picks the range [y * nr, (y + 1) * nr)); flags bit 0: no phases (every wave of a workgroup starts its sweep in step 0)
KARG = dict(A=0, lda=8, B=16, ldb=24, bias=32, res=40, nr=48, flags=52, C=56, ldc=64, cmask=72, N=76, alpha=80, npanels=84, bits=88,
            key=96, thr=100, scale=104, row_mult=108, seed_dev=112, stream_key=120, grid=124)
KARG_BYTES = 128

# ---- SGPRs (s0..s3 are free after the prologue: timing builds use them)
S_A, S_LDA, S_B, S_LDB, S_BIAS, S_RES, S_LDR, S_C = s(4, 2), s(6, 2), s(8, 2), s(10, 2), s(12, 2), s(14, 2), s(16, 2), s(18, 2)
S_LDC, S_CMASK, S_N, S_ALPHA, S_NPANELS, S_BITS = s(20, 2), s(22), s(23), s(24), s(25), s(26, 2)
S_KEY, S_THR, S_SCALE, S_ROWMULT, S_SEEDDEV, S_STREAMKEY, S_GRID = s(28), s(29), s(30), s(31), s(32, 2), s(34), s(35)
S_WID = s(36)
S_NLO = S_RES.sub(0)                  # first column of this workgroup's n-range (prologue only; the residual pointer's slot: no residual flavour exists)
S_NR, S_FLAGS = S_LDR.sub(0), S_LDR.sub(1)      # kernargs nr / flags (prologue only); afterwards s16:17 = exec mask of the 48 DMA lanes of a 768-byte W row (K = 384)
S_LO48 = s(16, 2)
S_NFULL, S_PHSTEP = s(46), s(47)      # N of the whole problem (sign-bit / dropout indexing; S_N = nr from the prologue on); steps between the phases of consecutive waves
S_LDA2, S_LDC2, S_LDR2 = s(37), s(38), s(39)
SRD_X, SRD_C, SRD_T = s(40, 4), s(48, 4), s(56, 4)
S_DUM, S_NEXTN0 = s(44), s(45)        # dummy steps left; n0 of the next step
S_P, S_PN = s(60), s(61)              # panel being accumulated; panel whose A is being fetched
S_LOOP = s(62)
S_WBASE_HI, S_LDB2, S_LDB2X48, S_WBASE = s(63), s(64), s(65), s(66)      # S_WBASE(_HI): B + w * 16 * ldb2
S_M0NEXT = s(67)                      # LDS address of this wave's first row in the buffer the NEXT tile is fetched into
S_NE2 = s(68)                         # 2 * n0 of the step whose epilogue is running
S_N0 = s(69)                          # n0 of the step being accumulated
S_CROW = [s(70 + i) for i in range(8)]   # store row offsets (w*64 + mb*32 + 8 it) * ldc2
S_XROW = s(78)                        # w * 64 * lda2
S_NB4 = s(79)                         # 4 * n0 of the next step (bias table offset)
S_STMASK = s(80, 2)                   # exec mask of the epilogue stores (0 until this wave's first n-step has been computed)
S_T = [s(82 + i) for i in range(8)]   # scratch
S_WPTR = s(90, 2)                     # address of the next W row this wave fetches
SRD_B = s(92, 4)                      # sign-bit buffer of the panel being stored (bits_out) / accumulated (bits_in)
S_BROW = [s(96), s(97)]               # (w*2 + mb) * N * 4: byte offset of the wave's slab row-block in the panel's sign bits
S_RN2, S_PPE = s(98), s(99)           # row_mult * N / 2 (dropout pairs per row); pair index of row 0 of the panel whose epilogue is running
S_ONE1, S_THR1, S_C1, S_RN32 = s(52), s(53), s(54), s(55)      # 0x00010001; (thr - 1) in both halves; 0x9E3779B1; 32 * S_RN2
S_LO32 = s(100, 2)                    # exec mask of lanes 0..31
N_SGPR = 102


# ---- VGPRs
def ACC(st, nb, mb):
    return v(st * 64 + (nb * 2 + mb) * 16, 16)


def BIASR(nb):
    return v(128 + nb * 16, 16)


def WFRAG(ks, nb):
    return v(160 + ((ks % 4) * 2 + nb) * 4, 4)


V_WRD = [v(192 + j) for j in range(8)]
V_LANE16, V_DMATMP = v(200), v(201)
V_COFF, V_CSTEP, V_STW, V_STRD, V_BIASRD, V_BIASSTEP = v(202), v(203), v(204), v(205), v(206), v(207)
V_XOFF = [v(208), v(209)]
V_STWX = v(210)                       # staging write address of the current 8-byte piece
V_PK = [v(212 + 2 * i, 2) for i in range(4)]       # converted pieces (rotating: a ds_write has read its data long before the pair comes round again)
V_RB = [[v(220 + 4 * i, 4) for i in range(4)]] * 2      # read-back (row-major) pieces = store data (slab 0's stores are issued before slab 1's reads)
V_L0, V_C8, V_H4 = v(236), v(237), v(211)           # flavour lane constants: dropout pair index of (row, 4 h); 8 * (lane & 31); 4 * (lane >> 5)
V_F = [v(238 + i) for i in range(18)]               # flavour temporaries (v244.. double as prologue temporaries)
V_TMP = [v(244 + i) for i in range(12)]
V_BCH = [v(220 + i) for i in range(16)]             # prologue: the bias chunks in flight (the read-back registers' space)


# ---- GELU flavour (no dropout / sign bits: their SGPRs and temporaries are free).  Constants: SGPR pairs, two fp32 values each, picked by op_sel / op_sel_hi
SG_PAIRS = [s(28, 2), s(30, 2), s(32, 2), s(52, 2), s(54, 2), s(92, 2), s(94, 2)]
def SGC(i):
    """(SGPR pair, broadcast selector) of GELU constant i: 0..9 = c0..c9, 10 = sqrt(zscale), 11 = -1, 12 = 1/2, 13 = clamp"""
    return SG_PAIRS[i // 2], (i % 2, i % 2)
S_GCLAMP = s(95)                      # = constant 13 as a single register (v_med3_f32)
G_T, G_Z, G_P = [v(238, 2), v(244, 2)], [v(240, 2), v(246, 2)], [v(242, 2), v(248, 2)]      # two interleaved chains
G_C9 = v(250, 2)                      # c9 in both halves (the Horner seed: a second SGPR operand would exceed the constant bus)


def BW(par, mb):
    """ReLU sign-bit words of a 32-row x 64-column slab as loaded (flavour bits_in has no bias: the bias registers' space)"""
    return v(128 + (par * 2 + mb) * 2, 2)


def XFRAG(mb, ks, KS=KS_DEFAULT):
    return a((mb * KS + ks) * 4, 4)


class QModel:
    """In-order completion queue (VM or LGKM) as the generator sees it: tags of the operations issued and not yet known retired."""

    def __init__(self, maxcnt):
        self.q = []
        self.maxcnt = maxcnt

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tags):
        """largest count n such that waiting for 'at most n outstanding' retires every tag in tags; None if none of them is in the queue"""
        idx = [i for i, t in enumerate(self.q) if t in tags]
        if not idx:
            return None
        return min(len(self.q) - 1 - max(idx), self.maxcnt)

    def wait(self, n):
        self.q = self.q[len(self.q) - n:] if n > 0 else []


class NtAsGen:
    CAP = 3            # filler groups per MFMA gap taken from the streams (fixed-slot instructions come on top)
    PF_GAP = 24        # bits_in: the sign-bit words of this step's slabs are requested here (consumed ~100 gaps later)
    DMA_END = 72       # the last LDS-DMA piece of the next tile is issued by this gap (~1.5 k cycles before the barrier)
    BAR_GAP = 119      # the barrier follows MFMA 119 (k-step 29); the last fragment reads of the tile are issued at gaps 113 / 115

    def __init__(self, name="svla_nt_as_f0", relu=False, drop=False, bits_out=False, bits_in=False, cap=None, dbg="", stagger=0, epi_order=1, dma_end=None, store_nt=True, load_nt=False, xstart=0, wphases=4, xburst=2, K=512, gelu=False):
        self.name = name
        self.gelu = gelu          # epilogue = erf-GELU(acc + bias) as the polynomial of asmgen/gelu_poly.py (the frozen ViT's fc1)
        assert not (gelu and (relu or drop or bits_out or bits_in))
        assert K in (384, 512)
        self.K, self.KS = K, K // 16
        if K != 512:      # the gap positions of the K = 512 schedule scaled to NG = 4 KS gaps
            self.BAR_GAP = 4 * self.KS - 9
            self.DMA_END = (72 * self.KS) // 32
        # epilogue flavour: relu (+ bits_out: the output's sign bits, + drop: train-mode dropout after the activation) | bits_in: alpha * product,
        # zeroed where the ReLU sign bit of the forward activation is 0 (no bias) | none of them: + bias
        self.relu, self.drop, self.bits_out, self.bits_in = relu, drop, bits_out, bits_in
        self.bias = not bits_in
        assert not (bits_in and (relu or drop or bits_out)) and (not bits_out or relu) and (not drop or relu)
        self.dbg = set(dbg.split(",")) if dbg else set()      # timing-only / bisection builds (tools/): time, nostore, nodma, nox, noepi, ...
        if cap is not None:
            self.CAP = cap
        self.stagger, self.epi_order, self.store_nt, self.load_nt, self.xstart = stagger, epi_order, store_nt, load_nt, xstart
        # panel switch: the 4 fragment loads (k-steps 4j .. 4j+3 of one row block) that share a 128-byte line of A are issued back to back (2; 1: row blocks
        # interleaved, 3: two lines per burst, 4: row blocks half a period apart, 0: each load as soon as its register is free).  Measured (profiles/
        # r04_nt_as_step_timing.txt): 2 = 4 > 3 > 0 > 1 -- a fragment-shaped load touches 32 bytes of 32 lines, and the L1 does not keep a line for the 600 cycles
        # until the next k-step's load comes for its next 32 bytes
        self.xburst = xburst
        self.wphases = wphases      # 4: every wave of a workgroup switches panels in its own step; 2: in pairs; 1: all in the same step
        if dma_end is not None:
            self.DMA_END = dma_end
        self.p = Prog(name)
        self.vm = QModel(63)
        self.lg = QModel(15)
        self.uid = 0
        self.stats = {}

    # ------------------------------------------------------------------ helpers
    def tag(self, base):
        self.uid += 1
        return f"{base}#{self.uid}"

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
        # the LGKM counter has 4 bits: never let the model's queue pass 15 (LDS operations retire in order within ~100 cycles, so in
        # steady state this wait finds its operations long retired)
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
        self.lg.issue(self.tag("dsw"))

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        T = V_TMP
        p.s_load(s(4, 16), s(0, 2), 0)
        p.s_load(s(20, 16), s(0, 2), 64)
        p.v_and_b32(T[0], 63, v(0))                 # lane
        p.v_lshrrev_b32(T[1], 6, v(0))              # wave
        p.v_readfirstlane_b32(S_WID, T[1])
        p.v_and_b32(T[2], 31, T[0])                 # c = lane & 31
        p.v_lshrrev_b32(T[3], 5, T[0])              # h = lane >> 5
        p.s_waitcnt(lgkmcnt=0)
        p.s_mov_b64(S_STMASK, 0)
        # ---- this workgroup's n-range: from here on S_N is the width of the range it sweeps, pointers / offsets are moved to its first column
        p.s_mov_b32(S_NFULL, S_N)
        p.s_mov_b32(S_N, S_NR)
        p.s_mul_i32(S_NLO, s(3), S_NR)              # workgroup_id_y * nr
        if "time" in self.dbg:
            for i in range(28, 35):
                p.s_mov_b32(s(i), 0)
        p.s_lshl_b32(S_LDA2, S_LDA.sub(0), 1)
        p.s_lshl_b32(S_LDC2, S_LDC.sub(0), 1)
        p.s_lshl_b32(S_LDR2, S_LDR.sub(0), 1)
        p.s_lshl_b32(S_LDB2, S_LDB.sub(0), 1)
        p.s_mul_i32(S_LDB2X48, S_LDB2, 48)
        p.s_mul_i32(S_T[0], S_NLO, S_LDB2)          # (< 2^32: N <= 4096 rows)
        p.s_add_u32(S_B.sub(0), S_B.sub(0), S_T[0])
        p.s_addc_u32(S_B.sub(1), S_B.sub(1), 0)
        p.s_lshl_b32(S_T[0], S_NLO, 2)
        p.s_add_u32(S_BIAS.sub(0), S_BIAS.sub(0), S_T[0])
        p.s_addc_u32(S_BIAS.sub(1), S_BIAS.sub(1), 0)
        p.s_lshl_b32(S_T[0], S_WID, 4)
        p.s_mul_i32(S_T[1], S_T[0], S_LDB2)         # w * 16 * ldb2 (< 2^32: N <= 4096 rows)
        p.s_add_u32(S_WBASE, S_B.sub(0), S_T[1])
        p.s_addc_u32(S_WBASE_HI, S_B.sub(1), 0)
        p.s_lshl_b32(S_M0NEXT, S_WID, 14)           # w * 16 KiB: the prologue fetches tile 0 into buffer 0
        p.s_lshl_b32(S_T[0], S_WID, 6)
        p.s_mul_i32(S_XROW, S_T[0], S_LDA2)         # w * 64 * lda2
        for mb in range(2):
            for it in range(4):
                p.s_add_u32(S_T[1], S_T[0], mb * 32 + 8 * it)
                p.s_mul_i32(S_CROW[mb * 4 + it], S_T[1], S_LDC2)
        # descriptors: bias table source; A / C / residual per panel
        p.s_mov_b32(SRD_T.sub(0), S_BIAS.sub(0))
        p.s_and_b32(SRD_T.sub(1), S_BIAS.sub(1), 0xffff)
        p.s_mov_b32(SRD_T.sub(3), 0x00020000)
        for srd in (SRD_X, SRD_C, SRD_B):
            p.s_mov_b32(srd.sub(2), 0xffffffff)
            p.s_mov_b32(srd.sub(3), 0x00020000)
        # ---- lane constants
        # fragment read addresses: row c (1 KiB pitch), logical 16-byte chunk 2 j + h, physical chunk = logical ^ (c & 15)
        p.v_and_b32(T[4], 15, T[2])
        p.v_lshlrev_b32(T[5], 10, T[2])             # c * 1024
        for j in range(8):
            p.v_add_u32(T[6], 2 * j, T[3])
            p.v_xor_b32(T[6], T[6], T[4])
            p.v_lshl_add_u32(V_WRD[j], T[6], 4, T[5])
        p.v_lshlrev_b32(V_LANE16, 4, T[0])
        # C store: lane -> (row lane >> 3, 16-byte chunk lane & 7)
        p.v_lshrrev_b32(T[6], 3, T[0])              # lane >> 3
        p.v_and_b32(T[7], 7, T[0])                  # lane & 7
        p.v_mul_lo_u32(T[8], T[6], S_LDC2)
        p.v_lshl_add_u32(V_COFF, T[7], 4, T[8])
        p.s_lshl_b32(S_T[3], S_NLO, 1)
        p.v_add_u32(V_COFF, S_T[3], V_COFF)         # + 2 * n_lo bytes
        # staging (wave-private 4 KiB at LDS_STG + w * 4096): [32 rows][128 B], 16-byte chunk q of row r stored at chunk q ^ ((r >> 1) & 7), and in
        # rows 16-31 the two 8-byte halves of a chunk are swapped: the 32 lanes of an accumulator-layout write (8 bytes each, rows 0-31, one chunk) then
        # cover 32 different 8-byte slots of the 256 bytes the LDS serves per clock (with q ^ (r & 7), rows r, r + 8, r + 16, r + 24 met in one slot:
        # 13-15 % of the LDS cycles were bank conflicts).  Row-major side: row R = (lane >> 3) + 8 it -> (R >> 1) & 7 = ((lane >> 4) + 4 it) & 7: the
        # address of odd it is V_STRD ^ 64
        p.s_lshl_b32(S_T[2], S_WID, 12)
        p.s_add_u32(S_T[2], S_T[2], LDS_STG)
        p.v_lshrrev_b32(T[8], 1, T[2])
        p.v_and_b32(T[8], 7, T[8])                  # (c >> 1) & 7
        p.v_lshlrev_b32(T[9], 7, T[2])              # c * 128
        p.v_lshl_add_u32(T[9], T[8], 4, T[9])
        p.v_lshrrev_b32(T[8], 4, T[2])              # c >> 4
        p.v_xor_b32(T[8], T[8], T[3])               # half: h ^ (c >> 4)
        p.v_lshl_add_u32(T[9], T[8], 3, T[9])
        p.v_add_u32(V_STW, S_T[2], T[9])
        p.v_lshrrev_b32(T[8], 4, T[0])              # lane >> 4
        p.v_xor_b32(T[8], T[7], T[8])               # (lane & 7) ^ (lane >> 4)
        p.v_lshlrev_b32(T[9], 7, T[6])
        p.v_lshl_add_u32(T[9], T[8], 4, T[9])
        p.v_add_u32(V_STRD, S_T[2], T[9])
        p.v_lshlrev_b32(T[9], 4, T[3])
        p.v_add_u32(V_BIASRD, LDS_BIAS, T[9])       # + h * 16
        # A fragment loads: row (mb * 32 + c) of the wave's 64, 16 bytes at k = 16 ks + 8 h
        for mb in range(2):
            p.v_add_u32(T[8], mb * 32, T[2])
            p.v_mul_lo_u32(T[8], T[8], S_LDA2)
            p.v_lshl_add_u32(V_XOFF[mb], T[3], 4, T[8])
        # ---- flavour constants
        p.s_mov_b32(S_LO32.sub(0), 0xffffffff)
        p.s_mov_b32(S_LO32.sub(1), 0)
        p.v_lshlrev_b32(V_H4, 2, T[3])              # 4 h
        p.v_lshlrev_b32(V_C8, 3, T[2])              # 8 c
        p.s_lshl_b32(S_T[0], S_WID, 1)
        for mb in range(2):
            p.s_add_u32(S_T[1], S_T[0], mb)
            p.s_mul_i32(S_T[1], S_T[1], S_NFULL)
            p.s_add_u32(S_T[1], S_T[1], S_NLO)
            p.s_lshl_b32(S_BROW[mb], S_T[1], 2)     # (w*2 + mb) * (N/64) * 256 + (n_lo/64) * 256
        if self.drop:
            p.s_mul_i32(S_RN2, S_ROWMULT, S_NFULL)
            p.s_lshr_b32(S_RN2, S_RN2, 1)
            p.s_lshl_b32(S_RN32, S_RN2, 5)
            p.s_mov_b32(S_ONE1, 0x00010001)
            p.s_sub_u32(S_T[1], S_THR, 1)
            p.s_mul_i32(S_THR1, S_T[1], S_ONE1)      # (thr - 1) in both halves (thr <= 0xffff)
            p.s_mov_b32(S_C1, 0x9E3779B1)
            # pair index of (row w*64 + c of the panel, column 4 h): ((w*64 + c) * row_mult * N + 4 h) / 2; the host guarantees < 2^32 pairs
            p.s_lshl_b32(S_T[1], S_WID, 6)
            p.v_add_u32(T[8], S_T[1], T[2])
            p.v_mul_lo_u32(T[8], T[8], S_RN2)
            p.v_lshl_add_u32(V_L0, T[3], 1, T[8])
            p.s_lshr_b32(S_T[1], S_NLO, 1)
            p.v_add_u32(V_L0, S_T[1], V_L0)         # + n_lo / 2 pairs
            # device-resident pass seed (recorded launch sequences): key = *seed_dev ^ stream_key
            p.s_or_b32(S_T[1], S_SEEDDEV.sub(0), S_SEEDDEV.sub(1))
            p.s_cmp("eq_u32", S_T[1], 0)
            p.s_cbranch_scc1("L_KEYOK")
            p.s_load(S_KEY, S_SEEDDEV, 0)
            p.s_waitcnt(lgkmcnt=0)
            p.s_xor_b32(S_KEY, S_KEY, S_STREAMKEY)
            p.label("L_KEYOK")
        elif self.bits_out:
            p.s_mov_b32(S_ONE1, 0x00010001)
        if self.gelu:
            import math as _m
            import struct as _st
            cs = gelu_poly.coefs() + [_st.unpack("<f", _st.pack("<f", _m.sqrt(gelu_poly.zscale())))[0], -1.0, 0.5, gelu_poly.CLAMP]
            for i, c in enumerate(cs):
                p.s_mov_b32(SG_PAIRS[i // 2].sub(i % 2), float(c))
        if self.bias:
            self.bias_loads()
        # ---- phase of this wave: phi = w * NS/4 + (workgroup & cmask) dummy steps before its first panel, (3 - w) * NS/4 after its last
        sh = {4: 0, 2: 1, 1: 2}[self.wphases]
        p.s_lshr_b32(S_T[0], S_N, 8 - sh)           # NS / wphases
        p.s_and_b32(S_T[1], S_FLAGS, 1)
        p.s_cmp("eq_u32", S_T[1], 0)
        p.s_cselect_b32(S_PHSTEP, S_T[0], 0)        # flags bit 0: no phases
        p.s_lshr_b32(S_T[1], S_WID, sh)             # phase index of this wave
        p.s_mul_i32(S_DUM, S_T[1], S_PHSTEP)
        if self.K == 384:
            p.s_mov_b32(S_LO48.sub(0), 0xffffffff)      # (nr / flags are consumed)
            p.s_mov_b32(S_LO48.sub(1), 0x0000ffff)
        if sh:
            p.s_lshl_b32(S_CMASK, S_CMASK, sh)      # the workgroups spread over NS / wphases steps
            p.s_or_b32(S_CMASK, S_CMASK, (1 << sh) - 1)
        p.s_and_b32(S_T[1], s(2), S_CMASK)
        p.s_add_u32(S_DUM, S_DUM, S_T[1])
        # ---- first panel
        p.s_mov_b32(S_P, s(2))
        p.s_mov_b32(S_PN, s(2))
        self.panel_srd(SRD_X, S_A, S_PN, S_LDA2)
        for ks in range(self.KS):
            for mb in range(2):
                if "nox" not in self.dbg:
                    p.buffer_load(XFRAG(mb, ks, self.KS), V_XOFF[mb], SRD_X, S_XROW, ks * 32)
            if ks == 23:
                p.s_waitcnt(vmcnt=0)            # (6-bit counter: bias chunks + 48 fragment loads so far; K = 384: the 16 DMA pieces follow)
        # ---- ROTATION: workgroup g starts the shared tile sequence at tile (g mod NS).  Without it every CU of the chip stores the same
        # 128-byte column of C (row stride 2 N bytes: one L2 / HBM channel group) and fetches the same W tile at the same moment.
        p.s_lshr_b32(S_T[0], S_N, 6)                # NS
        p.s_mov_b32(S_T[1], s(2))
        p.label("L_ROT")
        p.s_cmp("lt_u32", S_T[1], S_T[0])
        p.s_cbranch_scc1("L_ROTD")
        p.s_sub_u32(S_T[1], S_T[1], S_T[0])
        p.s_branch("L_ROT")
        p.label("L_ROTD")
        p.s_lshl_b32(S_N0, S_T[1], 6)               # n0 of the first tile
        p.s_mul_i32(S_T[2], S_N0, S_LDB2)           # byte offset of its first row (< 2^32)
        p.s_add_u32(S_WPTR.sub(0), S_WBASE, S_T[2])
        p.s_addc_u32(S_WPTR.sub(1), S_WBASE_HI, 0)
        for t in range(16):
            for grp in self.dma_groups(t):
                for th in grp:
                    th()
        p.s_xor_b32(S_M0NEXT, S_M0NEXT, 0x10000)
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        self.vm.wait(0)
        self.lg.wait(0)
        if self.bias:
            self.bias_table()
            p.s_waitcnt(lgkmcnt=0)
        if self.gelu:           # (v250:251 are prologue temporaries up to here)
            p.v_mov_b32(G_C9.sub(0), SG_PAIRS[4].sub(1))
            p.v_mov_b32(G_C9.sub(1), SG_PAIRS[4].sub(1))
        if "time" in self.dbg:
            p.s_memtime(s(0, 2))
            p.s_waitcnt(lgkmcnt=0)
            p.s_mov_b32(s(28), s(0))
        p.s_barrier()
        if self.bias:
            p.s_lshl_b32(S_T[2], S_N0, 2)
            p.v_add_u32(V_BIASSTEP, S_T[2], V_BIASRD)
            for nb in range(2):
                for rg in range(4):
                    p.ds_read(BIASR(nb).sub(4 * rg, 4), V_BIASSTEP, nb * 128 + rg * 32)
        for ks in range(3):
            for nb in range(2):
                self.wread(ks, nb)

    def bias_loads(self):
        """bias[N] -> VGPRs, every 1-KiB chunk requested before anything waits (the host always passes a bias pointer: zeros when the GEMM has none).  With
        one wait per chunk the prologue of a launch that runs a handful of n-steps per workgroup (mid-M) spent 8 serial memory latencies on a 2048-wide
        bias.  num_records = 4 N: reads past bias[N) return 0 -- the chunk offset is part of the VGPR offset (soffset is excluded from the hardware's range
        check: with it there, N % 256 == 128 read 512 bytes past the bias)."""
        p = self.p
        T = V_TMP
        p.v_lshlrev_b32(T[8], 2, v(0))
        p.s_lshl_b32(S_T[4], S_N, 2)
        p.s_mov_b32(SRD_T.sub(2), S_T[4])
        for i in range(16):       # N <= 4096
            if i:
                p.s_cmp("gt_u32", S_T[4], 1024 * i)
                p.s_cbranch_scc0("L_BIASLD")
            if "nobias" in self.dbg:
                p.v_mov_b32(V_BCH[i], 0)
            else:
                p.buffer_load(V_BCH[i], T[8], SRD_T, 0, 0)
            if i < 15:
                p.v_add_u32(T[8], 1024, T[8])
        p.label("L_BIASLD")

    def bias_table(self):
        """the chunks -> LDS (after the wait that also covers the first panel's A fragments and the first W tile)"""
        p = self.p
        T = V_TMP
        p.v_lshlrev_b32(T[8], 2, v(0))
        p.v_add_u32(T[9], LDS_BIAS, T[8])
        p.s_lshl_b32(S_T[4], S_N, 2)
        for i in range(16):
            if i:
                p.s_cmp("gt_u32", S_T[4], 1024 * i)
                p.s_cbranch_scc0("L_BIASST")
            p.ds_write(T[9], V_BCH[i], 0)
            if i == 7:
                p.s_waitcnt(lgkmcnt=0)          # (4-bit counter)
            if i < 15:
                p.v_add_u32(T[9], 1024, T[9])
        p.label("L_BIASST")

    def panel_srd(self, srd, base, panel, ld2):
        """srd.base = base + panel * 256 * ld2 (64-bit); one group: the carry travels through SCC"""
        p = self.p
        p.s_lshl_b32(S_T[5], panel, 8)
        p.s_mul_hi_u32(S_T[6], S_T[5], ld2)
        p.s_mul_i32(S_T[5], S_T[5], ld2)
        p.s_add_u32(srd.sub(0), base.sub(0), S_T[5])
        p.s_addc_u32(S_T[6], base.sub(1), S_T[6])
        p.s_and_b32(srd.sub(1), S_T[6], 0xffff)

    def bits_srd(self, panel):
        """SRD_B.base = bits + panel * 8 slab rows * (N/64) slabs * 256 B = bits + panel * N * 32"""
        p = self.p
        p.s_mul_i32(S_T[5], panel, S_NFULL)
        p.s_lshl_b32(S_T[5], S_T[5], 5)
        p.s_add_u32(SRD_B.sub(0), S_BITS.sub(0), S_T[5])
        p.s_addc_u32(S_T[6], S_BITS.sub(1), 0)
        p.s_and_b32(SRD_B.sub(1), S_T[6], 0xffff)

    def flavour_panel_scalars(self):
        """quantities of the panel whose epilogue runs from now on (set where SRD_C is)"""
        p = self.p
        if self.bits_out:
            self.bits_srd(S_P)
        if self.drop:
            p.s_lshl_b32(S_T[5], S_P, 8)
            p.s_mul_i32(S_PPE, S_T[5], S_RN2)

    def dma_groups(self, t):
        """LDS-DMA of row (w*16 + t) of the next W tile, as scheduler groups"""
        p = self.p

        def g1():
            if t == 0:
                p.s_mov_b32(M0, S_M0NEXT)
            else:
                p.s_add_u32(M0, M0, 1024)
            # row & 15 == t: lane i lands at physical chunk i, so it fetches logical chunk i ^ t
            p.v_xor_b32(V_DMATMP, t << 4, V_LANE16)

        def g2():
            if "nodma" not in self.dbg:
                if self.K == 384:      # a W row is 768 bytes = 48 lanes (LDS pitch stays 1 KiB): lanes 48..63 fetch nothing
                    p.s_mov_b64(EXEC, S_LO48)
                p.global_load_lds_x4(V_DMATMP, S_WPTR)
                if self.K == 384:
                    p.s_mov_b64(EXEC, -1)
                self.vm.issue(f"dma{t}")

        def g3():
            p.s_add_u32(S_WPTR.sub(0), S_WPTR.sub(0), S_LDB2)
            p.s_addc_u32(S_WPTR.sub(1), S_WPTR.sub(1), 0)
        return [[g1], [g2], [g3]]

    def wread(self, ks, nb):
        j = ks % 8
        self.ds_read(WFRAG(ks, nb), V_WRD[j], nb * 32768 + (ks // 8) * 256, f"w{ks}_{nb}")

    def step_head(self):
        """scalar bookkeeping every step starts with (one block, before its first MFMA): the next tile's n0 and W row pointer (the tile
        sequence wraps at N)"""
        p = self.p
        p.s_add_u32(S_NEXTN0, S_N0, 64)
        p.s_cmp("eq_u32", S_NEXTN0, S_N)
        p.s_cselect_b32(S_NEXTN0, 0, S_NEXTN0)
        p.s_lshl_b32(S_NB4, S_NEXTN0, 2)
        p.s_add_u32(S_WPTR.sub(0), S_WPTR.sub(0), S_LDB2X48)
        p.s_addc_u32(S_WPTR.sub(1), S_WPTR.sub(1), 0)
        p.s_cmp("eq_u32", S_NEXTN0, 0)
        p.s_cselect_b32(S_WPTR.sub(0), S_WBASE, S_WPTR.sub(0))
        p.s_cselect_b32(S_WPTR.sub(1), S_WBASE_HI, S_WPTR.sub(1))

    def barrier(self, kind):
        p = self.p
        nv = self.vm.need({f"dma{t}" for t in range(16)})
        if "time" in self.dbg:
            p.s_memtime(s(0, 2))
        if "nobarwait" in self.dbg:      # timing-only: no wait for the DMA at the barrier (wrong results)
            p.s_waitcnt(lgkmcnt=0)
        else:
            p.s_waitcnt(vmcnt=nv if nv is not None else 0, lgkmcnt=0)
        self.vm.wait(nv if nv is not None else 0)
        self.lg.wait(0)
        if "time" in self.dbg and kind != "dummy":
            ki = {"first": 0, "mid": 1, "last": 2}[kind]
            p.s_sub_u32(s(3), s(0), s(28))
            p.s_add_u32(s(29 + ki), s(29 + ki), s(3))
            p.s_add_u32(s(32 + ki), s(32 + ki), 1)
        if "time" in self.dbg:
            p.s_mov_b32(s(28), s(0))
        p.s_barrier()
        if self.stagger and kind != "dummy":
            # the four waves leave the barrier in the same cycle and run the same stream: every LDS read / DMA issue of the step would
            # collide with the other three waves' (measured: SQ_WAIT_INST_LDS = 1.5 quad-cycles per MFMA).  Wave w idles w * (stagger + ~3)
            # issue slots here, so the waves sit at different offsets inside the 32-cycle MFMA period until the next barrier.
            self.uid += 1
            done = f"L_STG{self.uid}"
            for i in range(1, 4):
                p.s_cmp("lt_u32", S_WID, i)
                p.s_cbranch_scc1(done)
                p.s_nop(self.stagger - 1)
            p.label(done)

    def step_tail_scalars(self):
        p = self.p
        p.s_lshl_b32(S_NE2, S_N0, 1)
        p.s_mov_b32(S_N0, S_NEXTN0)
        p.s_xor_b32(S_M0NEXT, S_M0NEXT, 0x10000)

    # ------------------------------------------------------------------ epilogue of one n-step (previous step's accumulators)
    def epi_stream(self, st, masked, earliest, late=None):
        """list of (earliest_gap, [thunks]): the epilogue of accumulator set st, stores optionally under S_STMASK.  Order (epi_order 1):
        convert + stage slab 0, read it back, convert + stage slab 1 (LDS operations of a wave execute in order: the writes follow the
        reads), store slab 0, read slab 1 back, [late: groups of the caller, e.g. the next step's bias reads], store slab 1 -- no store waits
        for a read-back issued just ahead of it."""
        p = self.p
        items = []

        def add(*ths, e=earliest):
            items.append((e, list(ths)))
        add(lambda: p.v_add_u32(V_CSTEP, S_NE2, V_COFF))

        C1 = 0x9E3779B1
        F = V_F
        VS, X, TT, MSK, A1, OBW = F[0:4], F[4], F[5], F[6], F[7], [F[8], F[9]]      # OBW: an even-aligned pair (one 8-byte store)
        U, U2 = F[10], F[11]
        BS = [F[12], F[13]]
        par = st          # accumulator set whose epilogue this is = parity of the sign-bit words prefetched for it

        def step_scalars():
            # per-step scalars of the epilogue's slabs: dropout pair offsets, sign-bit slab offsets
            if self.drop:
                p.s_lshr_b32(S_T[0], S_NE2, 2)
                p.s_add_u32(S_T[0], S_T[0], S_PPE)
                p.s_add_u32(S_T[1], S_T[0], S_RN32)
            if self.bits_out:
                p.s_lshl_b32(S_T[2], S_NE2, 1)
                p.s_add_u32(S_T[3], S_T[2], S_BROW[1])
                p.s_add_u32(S_T[2], S_T[2], S_BROW[0])
        if self.drop or self.bits_out:
            add(step_scalars)

        def convert(mb):
            if self.drop:
                # A1 = pair index of (this lane's row, column n0 + 4 h) times C1; pair c of the slab row: (A1 + c C1) ^ key -> drop_mix
                add(lambda mb=mb: p.v_add_u32(A1, S_T[mb], V_L0))
                add(lambda: p.v_mul_lo_u32(A1, A1, S_C1))
            if self.bits_in:
                def shift(mb=mb):
                    self.wait_for(vm_tags=[f"bw{par}_{mb}"])
                    for nb in range(2):
                        p.v_lshrrev_b32(BS[nb], V_H4, BW(par, mb).sub(nb))      # bit 8 rg + e = the sign bit of column nb*32 + 8 rg + 4 h + e
                add(shift)
            for nb in range(2):
                for rg in range(4):
                    gi = nb * 4 + rg
                    acc = ACC(st, nb, mb)
                    pk = V_PK[gi % 4]
                    if self.gelu:
                        # the two element pairs of the piece as two interleaved dependent chains (19 instructions each), emitted six instructions per group
                        ins = []
                        for ch in range(2):
                            gX, gT, gZ, gP = acc.sub(4 * rg + 2 * ch, 2), G_T[ch], G_Z[ch], G_P[ch]
                            seq = [lambda gX=gX, gT=gT: p.v_med3_f32(gT.sub(0), gX.sub(0), S_GCLAMP, S_GCLAMP, neg=(0, 1, 0)),
                                   lambda gX=gX, gT=gT: p.v_med3_f32(gT.sub(1), gX.sub(1), S_GCLAMP, S_GCLAMP, neg=(0, 1, 0)),
                                   lambda gT=gT, gZ=gZ: p.v_pk_mul_f32(gZ, gT, SGC(10)[0], sel=((0, 1), SGC(10)[1])),
                                   lambda gZ=gZ: p.v_pk_fma_f32(gZ, gZ, gZ, SGC(11)[0], sel=((0, 1), (0, 1), SGC(11)[1])),
                                   lambda gZ=gZ, gP=gP: p.v_pk_fma_f32(gP, gZ, G_C9, SGC(8)[0], sel=((0, 1), (0, 1), SGC(8)[1]))]
                            for k in range(7, -1, -1):
                                seq.append(lambda gZ=gZ, gP=gP, k=k: p.v_pk_fma_f32(gP, gP, gZ, SGC(k)[0], sel=((0, 1), (0, 1), SGC(k)[1])))
                            seq += [lambda gT=gT, gP=gP: p.v_pk_mul_f32(gP, gP, gT),
                                    lambda gP=gP: p.v_med3_f32(gP.sub(0), gP.sub(0), -0.5, 0.5),
                                    lambda gP=gP: p.v_med3_f32(gP.sub(1), gP.sub(1), -0.5, 0.5),
                                    lambda gP=gP: p.v_pk_add_f32(gP, gP, SGC(12)[0], sel=((0, 1), SGC(12)[1])),
                                    lambda gX=gX, gP=gP: p.v_pk_mul_f32(gP, gP, gX),
                                    lambda gP=gP, ch=ch, pk=pk: p.v_cvt_pk_bf16_f32(pk.sub(ch), gP.sub(0), gP.sub(1))]
                            ins.append(seq)
                        inter = [th for pair in zip(*ins) for th in pair]
                        for i0 in range(0, len(inter), 6):
                            grp = inter[i0:i0 + 6]
                            add(lambda grp=grp: [th() for th in grp])
                        add(lambda gi=gi: p.v_xor_b32(V_STWX, gi << 4, V_STW))
                        add(lambda pk=pk: self.ds_write(V_STWX, pk))
                        continue
                    for pr in range(2):
                        src = [acc.sub(4 * rg + 2 * pr), acc.sub(4 * rg + 2 * pr + 1)]
                        if self.bits_in:
                            def masked_pair(src=src, pr=pr, nb=nb, rg=rg, pk=pk):
                                for e in range(2):
                                    p.v_bfe_i32(TT, BS[nb], 8 * rg + 2 * pr + e, 1)
                                    p.v_mul_f32(VS[e], S_ALPHA, src[e])
                                    p.v_and_b32(VS[e], TT, VS[e])
                                p.v_cvt_pk_bf16_f32(pk.sub(pr), VS[0], VS[1])
                            add(masked_pair)
                            continue
                        if self.drop:
                            c = nb * 16 + rg * 4 + pr

                            def hash_pair(c=c):
                                p.v_add_u32(X, (c * C1) & 0xffffffff, A1)
                                p.v_xor_b32(X, S_KEY, X)
                                p.v_lshrrev_b32(TT, 16, X)
                                p.v_xor_b32(X, TT, X)
                                p.v_mul_u32_u24(X, 0xEB352D, X)
                                p.v_lshrrev_b32(TT, 13, X)
                                p.v_xor_b32(X, TT, X)
                                p.v_mul_u32_u24(X, 0x6CA68B, X)
                                p.v_lshrrev_b32(TT, 16, X)
                                p.v_xor_b32(X, TT, X)
                            add(hash_pair)

                            def keep_mask():
                                # per 16-bit half: all ones where the half >= thr (kept): sat(half - (thr - 1)) != 0
                                p.v_pk_sub_u16(MSK, X, S_THR1, clamp=True)
                                p.v_pk_min_u16(MSK, MSK, S_ONE1)
                                p.v_pk_sub_u16(MSK, 0, MSK)
                            add(keep_mask)

                            def scaled(src=src, pr=pr, pk=pk):
                                p.v_mul_f32(VS[0], S_SCALE, src[0])
                                p.v_mul_f32(VS[1], S_SCALE, src[1])
                                p.v_cvt_pk_bf16_f32(pk.sub(pr), VS[0], VS[1])
                            add(scaled)
                            add(lambda pk=pk, pr=pr: (p.v_pk_max_i16(pk.sub(pr), pk.sub(pr), 0), p.v_and_b32(pk.sub(pr), MSK, pk.sub(pr))))
                        else:
                            add(lambda src=src, pk=pk, pr=pr: p.v_cvt_pk_bf16_f32(pk.sub(pr), src[0], src[1]))
                            if self.relu:
                                add(lambda pk=pk, pr=pr: p.v_pk_max_i16(pk.sub(pr), pk.sub(pr), 0))      # negative halves (and -0) -> +0
                    if self.bits_out:
                        def signbits(pk=pk, nb=nb, rg=rg):
                            # outputs are >= +0: min(half, 1) per half, the four bits folded into a nibble at bit 8 rg of the slab row's word
                            p.v_pk_min_u16(U, pk.sub(0), S_ONE1)
                            p.v_pk_min_u16(U2, pk.sub(1), S_ONE1)
                            p.v_lshl_or_b32(U, U2, 2, U)
                            p.v_lshrrev_b32(U2, 15, U)
                            p.v_or_b32(U, U2, U)
                            p.v_and_b32(U, 15, U)
                            if rg == 0:
                                p.v_mov_b32(OBW[nb], U)
                            else:
                                p.v_lshl_or_b32(OBW[nb], U, 8 * rg, OBW[nb])
                        add(signbits)
                    add(lambda gi=gi: p.v_xor_b32(V_STWX, gi << 4, V_STW))
                    add(lambda pk=pk: self.ds_write(V_STWX, pk))
            if self.bits_out:
                def bits_store(mb=mb):
                    # lanes c and c + 32 hold the two interleaved nibble sets of row c: shift into place, merge, ONE 8-byte store per row
                    for nb in range(2):
                        p.v_lshlrev_b32(OBW[nb], V_H4, OBW[nb])
                        p.v_mov_b32(U, OBW[nb])
                        p.v_permlane32_swap(OBW[nb], U)
                        p.v_or_b32(OBW[nb], U, OBW[nb])      # lanes 0..31: own | partner's
                    if "nostore" in self.dbg:
                        return
                    if masked:
                        p.s_and_b64(EXEC, S_STMASK, S_LO32)
                    else:
                        p.s_mov_b64(EXEC, S_LO32)
                    p.buffer_store(v(OBW[0].idx, 2), V_C8, SRD_B, S_T[2 + mb])
                    self.vm.issue(self.tag("stb"))
                    p.s_mov_b64(EXEC, -1)
                add(bits_store)

        def readback(mb):
            tags = []
            for it in range(4):
                tg = self.tag(f"rb{mb}_{it}")
                tags.append(tg)
                def rb(it=it, tg=tg, mb=mb):
                    if it & 1:
                        p.v_xor_b32(V_F[16], 64, V_STRD)
                    self.ds_read(V_RB[mb][it], V_F[16] if it & 1 else V_STRD, it * 1024, tg)
                add(rb)
            return tags

        def stores(mb, tags):
            for it in range(4):
                def st_(it=it, mb=mb, tags=tags):
                    self.wait_for(lg_tags=[tags[it]])
                    if it >= 2:         # rows 16-31 come back with the halves of every chunk swapped
                        p.v_swap_b32(V_RB[mb][it].sub(0), V_RB[mb][it].sub(2))
                        p.v_swap_b32(V_RB[mb][it].sub(1), V_RB[mb][it].sub(3))
                    if "nostore" in self.dbg:
                        return
                    if masked:
                        p.s_mov_b64(EXEC, S_STMASK)
                    p.buffer_store(V_RB[mb][it], V_CSTEP, SRD_C, S_CROW[mb * 4 + it], nt=self.store_nt)
                    self.vm.issue(self.tag("st"))
                    if masked:
                        p.s_mov_b64(EXEC, -1)
                add(st_)
        if self.epi_order == 0:
            for mb in range(2):
                convert(mb)
                stores(mb, readback(mb))
            items += late or []
        else:
            convert(0)
            t0 = readback(0)
            convert(1)
            stores(0, t0)
            t1 = readback(1)
            items += late or []
            stores(1, t1)
        return items

    # ------------------------------------------------------------------ one n-step body
    def body(self, kind, st):
        """kind: 'first' | 'mid' | 'last' of this wave's sweep over a panel; st: accumulator set of this step"""
        p = self.p
        KS = self.KS
        NG = 4 * KS
        BAR = self.BAR_GAP
        fixed = [[] for _ in range(NG)]
        pre = [[] for _ in range(NG)]
        streams = []

        # ---- W fragment reads, ring of 4 k-steps: (ks + 3, nb) right after the MFMA (ks, nb, mb = 0) -- its ring slot was last read by the
        # MFMAs of k-step ks - 1.  The first three k-steps of the NEXT tile are read behind the barrier, at gaps BAR + 1 .. BAR + 6.
        for ks in range(KS - 3):
            for nb in range(2):
                fixed[4 * ks + 2 * nb + 1].append(lambda k3=ks + 3, nb=nb: self.wread(k3, nb))
        for k in range(3):
            for nb in range(2):
                fixed[BAR + 1 + 2 * k + nb].append(lambda k=k, nb=nb: self.wread(k, nb))
        # toggle the read addresses to the other buffer after their last use for this tile (chunk class j: last k-step 24 + j, read at gaps 4 (21 + j) + 1 / + 3)
        for j in range(8):
            fixed[4 * (KS - 11 + j) + 3].append(lambda j=j: p.v_xor_b32(V_WRD[j], 0x10000, V_WRD[j]))
        # ---- barrier: every read of this tile issued (last: k-step 31 at gaps 113 / 115) and retired, own DMA pieces of the next tile landed
        fixed[BAR].insert(0, lambda: self.barrier(kind))

        # ---- LDS-DMA of the next tile
        self.step_head()
        if self.bits_in:
            if kind == "first":
                self.bits_srd(S_P)
            # sign-bit words of this step's two slabs (consumed by its epilogue, in the next step): 8 bytes per lane = the slab row's word
            def prefetch_bits():
                for mb in range(2):
                    p.s_lshl_b32(S_T[4], S_N0, 2)
                    p.s_add_u32(S_T[4], S_T[4], S_BROW[mb])
                    p.buffer_load(BW(st, mb), V_C8, SRD_B, S_T[4])
                    self.vm.issue(f"bw{st}_{mb}")
            fixed[self.PF_GAP].append(prefetch_bits)
        dma = []
        for t in range(16):
            for grp in self.dma_groups(t):
                dma.append((1, grp))
        streams.append(dma)

        # ---- panel switch (last step of a panel): the next panel's A fragments replace the ones whose last MFMA has been issued
        if kind == "last":
            p.s_add_u32(S_PN, S_P, S_GRID)
            p.s_cmp("lt_u32", S_PN, S_NPANELS)
            # past the end (a mid-M workgroup's ONLY panel; every workgroup's last one): the fragment loads stay in the stream -- the counted waits of the
            # following code assume them -- but with num_records = 0 every lane is out of range, so they return zeros without a memory request.  (With the
            # real re-fetch of the last panel the last step of a one-panel workgroup took 11.9 k cycles instead of 3.6 k: profiles/r05_midm_step_timing.txt.)
            p.s_cselect_b32(SRD_X.sub(2), 0xffffffff, 0)
            p.s_sub_u32(S_T[7], S_NPANELS, 1)
            p.s_min_u32(S_PN, S_PN, S_T[7])
            self.panel_srd(SRD_X, S_A, S_PN, S_LDA2)
            for ks in range(KS):
                for mb in range(2):
                    def xl(ks=ks, mb=mb):
                        if "nox" in self.dbg:
                            return
                        p.buffer_load(XFRAG(mb, ks, self.KS), V_XOFF[mb], SRD_X, S_XROW, ks * 32, nt=self.load_nt)
                        self.vm.issue(f"x{ks}")
                    if self.xburst:
                        continue
                    fixed[max(4 * ks + 3, self.xstart + (2 * ks + mb) // 2 if self.xstart else 0)].append(xl)
            if self.xburst:
                def xl2(ks, mb):
                    def f():
                        if "nox" in self.dbg:
                            return
                        p.buffer_load(XFRAG(mb, ks, self.KS), V_XOFF[mb], SRD_X, S_XROW, ks * 32, nt=self.load_nt)
                        self.vm.issue(f"x{ks}")
                    return f
                if self.xburst == 3:          # two lines (8 k-steps) per burst
                    for j in range(KS // 8):
                        for mb in range(2):
                            for ks in range(8 * j, 8 * j + 8):
                                fixed[32 * j + 31].append(xl2(ks, mb))
                elif self.xburst == 4:        # the two row blocks half a line period apart
                    for j in range(KS // 4):
                        for mb in range(2):
                            for ks in range(4 * j, 4 * j + 4):
                                fixed[min(16 * j + 15 + 8 * mb, 4 * KS - 1)].append(xl2(ks, mb))
                else:
                    for j in range(KS // 4):
                        order = [(ks, mb) for ks in range(4 * j, 4 * j + 4) for mb in range(2)] if self.xburst == 1 else \
                                [(ks, mb) for mb in range(2) for ks in range(4 * j, 4 * j + 4)]
                        for ks, mb in order:
                            fixed[16 * j + 15].append(xl2(ks, mb))
        if kind == "first":
            for ks in range(KS):
                pre[4 * ks].append(("vm", f"x{ks}"))

        # ---- epilogue of the previous step, bias registers of the next step, then the bookkeeping that must follow the epilogue
        late = []
        if self.bias:
            late.append((16, [lambda: p.v_add_u32(V_BIASSTEP, S_NB4, V_BIASRD)]))
            for nb in range(2):
                for rg in range(4):
                    late.append((16, [lambda nb=nb, rg=rg: self.ds_read(BIASR(nb).sub(4 * rg, 4), V_BIASSTEP, nb * 128 + rg * 32, self.tag("bias"))]))
        epi = self.epi_stream(st ^ 1, masked=(kind == "first"), earliest=8, late=late) if "noepi" not in self.dbg else late
        epi.append((16, [self.step_tail_scalars]))
        if kind == "first":
            def first_tail():
                p.s_mov_b64(S_STMASK, -1)
                self.panel_srd(SRD_C, S_C, S_P, S_LDC2)
                self.flavour_panel_scalars()
                p.s_lshr_b32(S_LOOP, S_N, 7)
                p.s_sub_u32(S_LOOP, S_LOOP, 1)
            epi.append((16, [first_tail]))
        streams.append(epi)

        # ---- emit.  Each stream is PACED over its window of gaps (first gap, last gap): group k of n is due at first + k * span / n, so
        # the DMA pieces, the stores and the epilogue's LDS traffic leave the wave as an even trickle instead of a burst at the head of the step
        # (the CU's L1 <-> L2 path is the resource this kernel saturates: measured)
        windows = [(1, self.DMA_END), (8, BAR - 1)]
        pos = [0] * len(streams)
        nfill = 0
        for g in range(NG):
            ks, i = divmod(g, 4)
            nb, mb = divmod(i, 2)
            lg_tags = [f"w{ks}_{nb}"]
            vm_tags = [t for (q, t) in pre[g] if q == "vm"]
            if ks == 0:
                lg_tags += [t for t in self.lg.q if t.startswith("bias")]
            self.wait_for(vm_tags=vm_tags, lg_tags=lg_tags)
            c = (BIASR(nb) if self.bias else 0) if ks == 0 else ACC(st, nb, mb)
            p.v_mfma_f32_32x32x16_bf16(ACC(st, nb, mb), WFRAG(ks, nb), XFRAG(mb, ks, self.KS), c)
            for th in fixed[g]:
                th()
            if g < BAR:
                for si, sm in enumerate(streams):
                    g0, g1 = windows[si]
                    due = len(sm) if g >= g1 else (0 if g < g0 else (len(sm) * (g - g0 + 1) + (g1 - g0)) // (g1 - g0 + 1))
                    n = 0
                    while pos[si] < due and sm[pos[si]][0] <= g and n < self.CAP:
                        for th in sm[pos[si]][1]:
                            th()
                        pos[si] += 1
                        n += 1
                        nfill += 1
            if g == BAR - 1:
                # every DMA piece the barrier's vmcnt must cover and every LDS operation of the epilogue has been issued: the LGKM queue a
                # step starts with is the six fragment reads behind the barrier, whatever step preceded it
                for si, sm in enumerate(streams):
                    while pos[si] < len(sm):
                        for th in sm[pos[si]][1]:
                            th()
                        pos[si] += 1
                        nfill += 1
        self.stats[(kind, st)] = nfill

    def dummy_body(self):
        """a step of a wave that has no panel in flight (before its first / after its last): its share of the DMA, the barrier, the
        fragment reads behind it (never consumed: they keep the LGKM state every step starts with), no MFMA, no epilogue"""
        p = self.p
        p.s_waitcnt(lgkmcnt=0)          # the fragment / bias reads the previous step left in flight: nobody consumes them here
        self.lg.wait(0)
        self.step_head()
        for t in range(16):
            for grp in self.dma_groups(t):
                for th in grp:
                    th()
        for j in range(8):
            p.v_xor_b32(V_WRD[j], 0x10000, V_WRD[j])
        if self.bias:
            p.v_add_u32(V_BIASSTEP, S_NB4, V_BIASRD)          # the next step may be this wave's first real one: its accumulator initialiser
            for nb in range(2):
                for rg in range(4):
                    self.ds_read(BIASR(nb).sub(4 * rg, 4), V_BIASSTEP, nb * 128 + rg * 32, self.tag("bias"))
        self.barrier("dummy")
        for k in range(3):
            for nb in range(2):
                self.wread(k, nb)
        self.step_tail_scalars()

    # ------------------------------------------------------------------ whole kernel
    def build(self):
        p = self.p
        self.prologue()
        p.label("L_DUMA")
        p.s_cmp("eq_u32", S_DUM, 0)
        p.s_cbranch_scc1("L_FIRST")
        self.dummy_body()
        p.s_sub_u32(S_DUM, S_DUM, 1)
        p.s_branch("L_DUMA")
        # the first generated body (LAST) is entered from a MID step at run time: seed the queue models with what such a step leaves behind
        real, self.p = self.p, Prog("scratch")
        self.body("mid", 0)
        self.p = real
        p.label("L_LAST")
        self.body("last", 1)
        p.s_add_u32(S_P, S_P, S_GRID)
        p.s_cmp("ge_u32", S_P, S_NPANELS)
        p.s_cbranch_scc1("L_DRAIN")
        p.label("L_FIRST")
        self.body("first", 0)
        p.s_cmp("eq_u32", S_LOOP, 0)
        p.s_cbranch_scc1("L_LAST")
        p.label("L_MID")
        self.body("mid", 1)
        self.body("mid", 0)
        p.s_sub_u32(S_LOOP, S_LOOP, 1)
        p.s_cmp("lg_u32", S_LOOP, 0)
        p.s_cbranch_scc1("L_MID")
        p.s_branch("L_LAST")
        p.label("L_DRAIN")
        # the last step's epilogue (accumulator set 1), nothing to hide it under; the fragments prefetched for a non-existent next step drain
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        self.vm.wait(0)
        self.lg.wait(0)
        for _, grp in self.epi_stream(1, masked=False, earliest=0):
            for th in grp:
                th()
        # trailing dummy steps: (3 - w) * NS/4, so that every wave of the workgroup passes the same number of barriers
        sh = {4: 0, 2: 1, 1: 2}[self.wphases]
        p.s_lshr_b32(S_T[1], S_WID, sh)
        p.s_sub_u32(S_T[1], self.wphases - 1, S_T[1])
        p.s_mul_i32(S_DUM, S_PHSTEP, S_T[1])
        p.label("L_DUMB")
        p.s_cmp("eq_u32", S_DUM, 0)
        p.s_cbranch_scc1("L_EXIT")
        self.dummy_body()
        p.s_sub_u32(S_DUM, S_DUM, 1)
        p.s_branch("L_DUMB")
        p.label("L_EXIT")
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        if "time" in self.dbg:
            # cycles between consecutive barriers, summed by the kind of step the interval ends in: [first, mid, last] sums, then counts, to
            # the debug buffer (kernarg 'bits') at ((workgroup * 4 + wave) * 8 + i) * 4
            p.s_mov_b32(SRD_T.sub(0), S_BITS.sub(0))
            p.s_and_b32(SRD_T.sub(1), S_BITS.sub(1), 0xffff)
            p.s_mov_b32(SRD_T.sub(2), 0xffffffff)
            p.s_lshl_b32(S_T[0], s(2), 2)
            p.s_add_u32(S_T[0], S_T[0], S_WID)
            p.s_lshl_b32(S_T[0], S_T[0], 5)
            p.v_mov_b32(V_TMP[0], 0)
            p.s_mov_b64(EXEC, 1)
            for i in range(6):
                p.v_mov_b32(V_TMP[1], s(29 + i))
                p.buffer_store(V_TMP[1], V_TMP[0], SRD_T, S_T[0], 4 * i)
            p.s_waitcnt(vmcnt=0)
        p.s_endpgm()
        return self

    # ------------------------------------------------------------------ assembly text
    def asm_text(self):
        body = self.p.text()
        name = self.name
        return f"""// GENERATED by safevla_amd/asmgen/nt_as_gen.py -- do not edit.
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.protected {name}
\t.globl {name}
\t.p2align 8
\t.type {name},@function
{name}:
{body}
.L{name}_end:
\t.size {name}, .L{name}_end-{name}

\t.rodata
\t.p2align 6
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {LDS_BYTES}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {KARG_BYTES}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 1
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 512
\t\t.amdhsa_next_free_sgpr {N_SGPR}
\t\t.amdhsa_accum_offset 256
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel

\t.amdgpu_metadata
---
amdhsa.version: [ 1, 2 ]
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.kernels:
  - .name: {name}
    .symbol: {name}.kd
    .kernarg_segment_size: {KARG_BYTES}
    .group_segment_fixed_size: {LDS_BYTES}
    .private_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .wavefront_size: 64
    .sgpr_count: {N_SGPR + 6}
    .vgpr_count: 512
    .agpr_count: 256
    .max_flat_workgroup_size: 256
    .args:
      - {{ .size: {KARG_BYTES}, .offset: 0, .value_kind: by_value }}
...
\t.end_amdgpu_metadata
"""


# flavour -> generator options (the C dispatcher nt_as_try of csrc/gemm.hip picks by name)
FLAVOURS = {
    "f0": dict(),                                               # bias (or none): in_proj forward, out_proj input gradient
    "f1d": dict(relu=True, bits_out=True, drop=True),           # bias, ReLU, dropout, sign bits out: linear1 forward in train mode
    "f1": dict(relu=True, bits_out=True),                       # ... eval mode / visual compressor
    "f3": dict(bits_in=True),                                   # alpha * product under the ReLU sign bits: input gradient through linear2 (+ dropout scale)
    "k384_f0": dict(K=384),                                     # K = 384 (24 k-steps): the frozen ViT-S/14's qkv projection (dino_preprocessors.py:27-35)
    "k384_f2": dict(K=384, gelu=True),                          # ... its fc1: bias + erf-GELU (asmgen/gelu_poly.py)
    "k384_f1": dict(K=384, relu=True, bits_out=True),           # the visual compressor's first layer on the 384-wide DINOv2 features (bias, ReLU, sign bits)
}
if _os.environ.get("SVLA_ASM_DEBUG_VARIANTS"):      # timing-only / bisection builds (tools/time_nt_as.py)
    for _d in ("time", "time,nostore", "time,nodma", "time,noepi", "time,nox", "time,nobarwait", "time,noepi,nodma,nox"):
        FLAVOURS["f0_" + _d.replace(",", "_")] = dict(dbg=_d)
    for _k, _o in (("xb0", dict(xburst=0)), ("xb3", dict(xburst=3)), ("wp1", dict(wphases=1)), ("wp2", dict(wphases=2))):
        FLAVOURS["f0_" + _k] = _o
    for _k in ("f1d", "f1", "f3"):
        FLAVOURS[_k + "_time"] = dict(FLAVOURS[_k], dbg="time") if _k != "f1d" else FLAVOURS[_k]
    for _d in ("time", "time,nostore", "time,nodma", "time,noepi", "time,noepi,nodma"):      # round 5: the K = 384 flavours (tools/time_midm.py)
        FLAVOURS["k384_f0_" + _d.replace(",", "_")] = dict(K=384, dbg=_d)


def generate(flavour="f0"):
    g = NtAsGen(name=f"svla_nt_as_{flavour}", **FLAVOURS[flavour])
    g.build()
    return g
