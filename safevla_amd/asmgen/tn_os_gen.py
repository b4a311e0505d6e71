"""Generator of the output-stationary weight-gradient GEMM for gfx950:  dW[N,K] (fp32) += sum_m dY[m,n] * X[m,k]  (+ db[n] += sum_m dY[m,n]).

Replaces gemm_tn8p_bf16_kernel of csrc/gemm.hip for the weight gradients of the fusion encoder's linears (reference: autograd of the
nn.TransformerEncoder layers, architecture/models/allenact_transformer_models/allenact_dino_transformer.py:545-552,702-708).  The reduction
runs over the (huge) row dimension, so the only stationary thing is the OUTPUT:

  * one workgroup of FOUR waves per CU (one per SIMD), each wave owns a 128 x 128 block of a 256 x 256 tile of dW: 16 MFMA 32x32
    accumulators = 256 AGPRs; the HIP kernel's 8 waves of 128 x 64 read 12 transposed fragments per 8 MFMAs, this one 16 per 16;
  * the row chunk of the workgroup streams through a ring of FOUR 32-row LDS slots (16 KiB of dY + 16 KiB of X each, row-major, 64-byte
    blocks XOR-swizzled by row & 3 on the DMA source address: the transposed reads are conflict free), LDS-DMA three slots ahead,
    ONE barrier per slot (32 MFMAs per wave), counted vmcnt(16);
  * fragments are gathered with ds_read_b64_tr_b16 (both operands are "transposed": the reduction index is the slow memory dimension),
    one k-step (16 rows) ahead, two register sets, one read per MFMA gap;
  * fp32 atomics from the AGPRs at the end of the chunk; the bias gradient rides on the dY fragments already in registers
    (v_dot2c_f32_bf16 against ones), on the waves / workgroups / slots whose turn it is.
Every wait is counted by the generator's queue models and checked by amdasm.Emu (tests/test_asm_emulator_cpu.py).
"""
from .amdasm import EXEC, M0, Prog, a, s, v
from .nt_as_gen import QModel

LDS_BYTES = 131072
SLOT = 32768               # bytes per ring slot: [32 rows][512 B] of dY, then the same of X
KARG = dict(dY=0, ldy=8, X=16, ldx=24, dW=32, ldw=40, db=48, M=56, N=60, K=64, chunk_rows=68, ntile=72, ntk=76, grid=80)
KARG_BYTES = 88

S_DY, S_LDY, S_X, S_LDX, S_DW, S_LDW, S_DB = s(4, 2), s(6, 2), s(8, 2), s(10, 2), s(12, 2), s(14, 2), s(16, 2)
S_M, S_N, S_K, S_CHUNKROWS, S_NTILE, S_NTK = s(18), s(19), s(20), s(21), s(22), s(23)
S_WID, S_WN, S_WK = s(24), s(25), s(26)
S_TILE, S_CHUNK, S_N0, S_K0, S_KT = s(27), s(28), s(29), s(30), s(31)
S_LDY2, S_LDX2, S_2LDY2, S_2LDX2 = s(32), s(33), s(34), s(35)
S_STEPA, S_STEPB = s(36, 2), s(38, 2)          # 32 rows * ld2 (64-bit); zeroed when the DMA cursor reaches the end of the chunk
S_PA, S_PB = s(40, 2), s(42, 2)                # next slot's first row of this wave's share: dY / X
S_TA, S_TB = s(44, 2), s(46, 2)                # running row pointers inside a slot
SRD_W, SRD_B = s(48, 4), s(52, 4)
S_LEFT, S_DLEFT = s(56), s(57)                 # slots left to compute / to fetch
S_M0W = s(58)                                  # w * 4096: this wave's 8 rows inside a slot part
S_BTURN, S_DOBIAS = s(59), s(60)
S_LDW4, S_ROW = s(61), s(62)
S_T = [s(64 + i) for i in range(12)]
N_SGPR = 80


def ACC(i, j):
    return a((i * 4 + j) * 16, 16)


def FA(f, u):
    return v(f * 32 + u * 4, 4)


def FB(f, u):
    return v(f * 32 + 16 + u * 4, 4)


V_RA = [[v(64 + h * 4 + u) for u in range(4)] for h in range(2)]       # transposed-read addresses of dY fragment u, slots 2h / 2h+1
V_RB = [[v(72 + h * 4 + u) for u in range(4)] for h in range(2)]
V_SA, V_SB = [v(80), v(81)], [v(82), v(83)]                           # DMA source offsets (row pair pattern t & 1)
V_BG = [v(84 + u) for u in range(4)]                                  # bias-gradient partial sums of dY fragment u's column
V_ONES, V_OFF, V_LANE = v(88), v(89), v(90)
V_T = [v(96 + i) for i in range(16)]


class TnOsGen:
    def __init__(self, name="svla_tn_os", dbg=""):
        self.name = name
        self.dbg = set(dbg.split(",")) if dbg else set()
        self.p = Prog(name)
        self.vm = QModel(63)
        self.lg = QModel(15)
        self.uid = 0

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

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        T = V_T
        p.s_load(s(4, 16), s(0, 2), 0)
        p.s_load(s(20, 4), s(0, 2), 64)
        p.v_and_b32(T[0], 63, v(0))                 # lane
        p.v_mov_b32(V_LANE, T[0])                   # (v0.. become fragment registers)
        p.v_lshrrev_b32(T[1], 6, v(0))
        p.v_readfirstlane_b32(S_WID, T[1])
        p.s_waitcnt(lgkmcnt=0)
        p.s_lshr_b32(S_WN, S_WID, 1)
        p.s_and_b32(S_WK, S_WID, 1)
        # ---- workgroup -> (tile, chunk): XCD-contiguous virtual index (workgroup b runs on XCD b % 8), tile = vid % ntile: the tiles that share
        # a row chunk share an L2
        p.s_load(S_T[8], s(0, 2), KARG["grid"])
        p.s_mov_b32(S_T[0], s(2))                   # workgroup id
        p.s_waitcnt(lgkmcnt=0)
        p.s_lshr_b32(S_T[1], S_T[8], 3)             # q = nwg / 8
        p.s_and_b32(S_T[2], S_T[8], 7)              # r = nwg % 8
        p.s_and_b32(S_T[3], S_T[0], 7)              # xcd
        p.s_lshr_b32(S_T[4], S_T[0], 3)             # slot
        p.s_add_u32(S_T[5], S_T[1], 1)              # q + 1
        p.s_cmp("lt_u32", S_T[3], S_T[2])
        p.s_cbranch_scc1("L_XLO")
        p.s_mul_i32(S_T[6], S_T[2], S_T[5])         # r * (q + 1)
        p.s_sub_u32(S_T[7], S_T[3], S_T[2])
        p.s_mul_i32(S_T[7], S_T[7], S_T[1])
        p.s_add_u32(S_T[6], S_T[6], S_T[7])
        p.s_branch("L_XD")
        p.label("L_XLO")
        p.s_mul_i32(S_T[6], S_T[3], S_T[5])
        p.label("L_XD")
        p.s_add_u32(S_T[6], S_T[6], S_T[4])         # vid
        p.s_mov_b32(S_CHUNK, 0)
        p.label("L_DIV")
        p.s_cmp("lt_u32", S_T[6], S_NTILE)
        p.s_cbranch_scc1("L_DIVD")
        p.s_sub_u32(S_T[6], S_T[6], S_NTILE)
        p.s_add_u32(S_CHUNK, S_CHUNK, 1)
        p.s_branch("L_DIV")
        p.label("L_DIVD")
        p.s_mov_b32(S_TILE, S_T[6])
        p.s_mov_b32(S_T[7], 0)                      # tile / ntk
        p.label("L_DIV2")
        p.s_cmp("lt_u32", S_T[6], S_NTK)
        p.s_cbranch_scc1("L_DIV2D")
        p.s_sub_u32(S_T[6], S_T[6], S_NTK)
        p.s_add_u32(S_T[7], S_T[7], 1)
        p.s_branch("L_DIV2")
        p.label("L_DIV2D")
        p.s_mov_b32(S_KT, S_T[6])
        p.s_lshl_b32(S_N0, S_T[7], 8)
        p.s_lshl_b32(S_K0, S_T[6], 8)
        # ---- rows of this chunk: [mbeg, mend), slots of 32
        p.s_mul_i32(S_T[0], S_CHUNK, S_CHUNKROWS)   # mbeg
        p.s_add_u32(S_T[1], S_T[0], S_CHUNKROWS)
        p.s_min_u32(S_T[1], S_T[1], S_M)            # mend
        p.s_sub_u32(S_T[1], S_T[1], S_T[0])
        p.s_lshr_b32(S_LEFT, S_T[1], 5)
        p.s_mov_b32(S_DLEFT, S_LEFT)
        p.s_cmp("eq_u32", S_LEFT, 0)
        p.s_cbranch_scc0("L_WORK")
        p.s_endpgm()
        p.label("L_WORK")
        p.s_lshl_b32(S_LDY2, S_LDY.sub(0), 1)
        p.s_lshl_b32(S_LDX2, S_LDX.sub(0), 1)
        p.s_lshl_b32(S_2LDY2, S_LDY2, 1)
        p.s_lshl_b32(S_2LDX2, S_LDX2, 1)
        p.s_lshl_b32(S_LDW4, S_LDW.sub(0), 2)
        for step, ld2 in ((S_STEPA, S_LDY2), (S_STEPB, S_LDX2)):
            p.s_lshl_b32(step.sub(0), ld2, 5)
            p.s_lshr_b32(step.sub(1), ld2, 27)
        # first row of this wave's share of slot 0: mbeg + 8 w; column block n0 / k0
        p.s_lshl_b32(S_T[2], S_WID, 3)
        p.s_add_u32(S_T[2], S_T[2], S_T[0])
        for ptr, base, ld2, col in ((S_PA, S_DY, S_LDY2, S_N0), (S_PB, S_X, S_LDX2, S_K0)):
            p.s_mul_hi_u32(S_T[4], S_T[2], ld2)
            p.s_mul_i32(S_T[3], S_T[2], ld2)
            p.s_lshl_b32(S_T[5], col, 1)
            p.s_add_u32(S_T[3], S_T[3], S_T[5])
            p.s_addc_u32(S_T[4], S_T[4], 0)
            p.s_add_u32(ptr.sub(0), base.sub(0), S_T[3])
            p.s_addc_u32(ptr.sub(1), base.sub(1), S_T[4])
        p.s_lshl_b32(S_M0W, S_WID, 12)
        # ---- lane constants
        p.v_and_b32(T[2], 31, T[0])                 # i & 31: 16-byte chunk inside the 512-byte row as it lands in LDS
        p.v_lshrrev_b32(T[3], 5, T[0])              # i >> 5: row of the pair
        p.v_lshrrev_b32(T[4], 2, T[2])              # physical 64-byte block
        p.v_and_b32(T[5], 3, T[2])
        for par in range(2):
            # rows 8 w + 2 t + (i >> 5), t & 1 = par: row & 3 = 2 par + (i >> 5); logical block = physical ^ (row & 3)
            p.v_add_u32(T[6], 2 * par, T[3])
            p.v_xor_b32(T[6], T[6], T[4])
            p.v_lshl_add_u32(T[6], T[6], 2, T[5])   # logical chunk
            p.v_lshlrev_b32(T[6], 4, T[6])
            p.v_mul_lo_u32(T[7], T[3], S_LDY2)
            p.v_add_u32(V_SA[par], T[6], T[7])
            p.v_mul_lo_u32(T[7], T[3], S_LDX2)
            p.v_add_u32(V_SB[par], T[6], T[7])
        # transposed reads: lane (q = l >> 4, pl = l & 15) addresses row 8 (q >> 1) + (pl >> 2) of the 16-row step, columns
        # col0 + 16 (q & 1) + 4 (pl & 3); the 64-byte block of fragment u of this wave = (wave's first block + u) ^ (row & 3)
        p.v_and_b32(T[2], 15, T[0])                 # pl
        p.v_lshrrev_b32(T[3], 4, T[0])              # q
        p.v_lshrrev_b32(T[4], 2, T[2])              # pl >> 2 = row & 3
        p.v_lshrrev_b32(T[5], 1, T[3])              # q >> 1
        p.v_lshl_add_u32(T[6], T[5], 3, T[4])       # row
        p.v_lshlrev_b32(T[6], 9, T[6])              # * 512
        p.v_and_b32(T[7], 1, T[3])
        p.v_lshlrev_b32(T[7], 5, T[7])              # 16 (q & 1) * 2 bytes
        p.v_and_b32(T[8], 3, T[2])
        p.v_lshl_add_u32(T[7], T[8], 3, T[7])       # + 4 (pl & 3) * 2
        p.v_add_u32(T[6], T[6], T[7])
        for part, regs, wsel in ((0, V_RA, S_WN), (16384, V_RB, S_WK)):
            p.s_lshl_b32(S_T[2], wsel, 2)           # wave's first 64-byte block: 128 columns = 4 blocks
            for u in range(4):
                p.s_add_u32(S_T[3], S_T[2], u)
                p.v_xor_b32(T[9], S_T[3], T[4])
                p.v_lshl_add_u32(T[9], T[9], 6, T[6])
                p.v_add_u32(regs[0][u], part, T[9])
                p.v_add_u32(regs[1][u], part + 2 * SLOT, T[9])
        p.v_mov_b32(V_ONES, 0x3f803f80)
        for u in range(4):
            p.v_mov_b32(V_BG[u], 0)
        for i in range(4):
            for j in range(4):
                for r in range(16):
                    p.v_accvgpr_write_b32(ACC(i, j).sub(r), 0)
        p.s_mov_b32(S_BTURN, S_KT)                  # slot x's bias gradient belongs to the workgroup whose k-tile index == x mod ntk
        # ---- ring fill: slots 0, 1, 2 and the dY part of slot 3 (the loop fetches the X part of slot x + 3 under k-step (x, 0) and the dY part of
        # slot x + 4 under (x, 1))
        for x, part in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0)):
            for grp in self.dma_groups(x, part):
                for th in grp:
                    th()
        self.wait_for(vm_tags=[f"dma{0}_{part}_{t}" for part in range(2) for t in range(4)])
        p.s_barrier()
        for u in range(4):
            self.tread(0, 0, 0, "a", u)
            self.tread(0, 0, 0, "b", u)

    # ------------------------------------------------------------------ pieces
    def dma_groups(self, x, part):
        """scheduler groups: the 4 LDS-DMA instructions of this wave for one part (0: dY, 1: X) of ring position x"""
        p = self.p
        run, cur, two, src = (S_TA, S_PA, S_2LDY2, V_SA) if part == 0 else (S_TB, S_PB, S_2LDX2, V_SB)
        step = S_STEPA if part == 0 else S_STEPB
        groups = []
        for t in range(4):
            def g(t=t):
                if t == 0:
                    p.s_mov_b32(run.sub(0), cur.sub(0))
                    p.s_mov_b32(run.sub(1), cur.sub(1))
                else:
                    p.s_add_u32(run.sub(0), run.sub(0), two)
                    p.s_addc_u32(run.sub(1), run.sub(1), 0)
                p.s_add_u32(M0, S_M0W, (x % 4) * SLOT + part * 16384 + t * 1024)
                p.s_nop(0)
                if "nodma" not in self.dbg:
                    p.global_load_lds_x4(src[t & 1], run)
                    self.vm.issue(f"dma{x % 4}_{part}_{t}")
                if t == 3:
                    if part == 1:
                        # advance the fetch cursor; past the end of the chunk it stays on the last slot (fetched again, never computed)
                        p.s_cmp("gt_u32", S_DLEFT, 1)              # S_DLEFT = slots from the cursor to the end of the chunk, the cursor's included
                        p.s_cselect_b32(S_T[8], S_STEPA.sub(0), 0)
                        p.s_cselect_b32(S_T[9], S_STEPA.sub(1), 0)
                        p.s_cselect_b32(S_T[10], S_STEPB.sub(0), 0)
                        p.s_cselect_b32(S_T[11], S_STEPB.sub(1), 0)
                        p.s_cselect_b32(S_T[7], 1, 0)
                        p.s_sub_u32(S_DLEFT, S_DLEFT, S_T[7])
                        p.s_add_u32(S_PA.sub(0), S_PA.sub(0), S_T[8])
                        p.s_addc_u32(S_PA.sub(1), S_PA.sub(1), S_T[9])
                        p.s_add_u32(S_PB.sub(0), S_PB.sub(0), S_T[10])
                        p.s_addc_u32(S_PB.sub(1), S_PB.sub(1), S_T[11])
            groups.append([g])
        return groups

    def tread(self, f, x, st, which, u):
        """the two transposed reads of fragment u of k-step st (0 / 1) of ring position x into register set f"""
        pos = x % 4
        regs = (V_RA if which == "a" else V_RB)[pos >> 1][u]
        frag = (FA if which == "a" else FB)(f, u)
        off = (pos & 1) * SLOT + st * 8192
        for hi in range(2):
            if len(self.lg.q) >= 15:
                self.p.s_waitcnt(lgkmcnt=11)
                self.lg.wait(11)
            self.p.ds_read_b64_tr_b16(frag.sub(2 * hi, 2), regs, off + hi * 2048)
            self.lg.issue(f"f{f}_{which}{u}")

    def kstep(self, x, st, last_of_slot_barrier):
        """16 MFMAs of k-step st of ring position x (register set st), with the next k-step's reads and a share of the DMA in their gaps"""
        p = self.p
        f = st
        fillers = []
        # next k-step: (x, 1) after (x, 0); (x + 1, 0) after (x, 1)
        nx, nst = (x, 1) if st == 0 else (x + 1, 0)
        for u in range(4):
            fillers.append(lambda u=u: self.tread(f ^ 1, nx, nst, "a", u))
            fillers.append(lambda u=u: self.tread(f ^ 1, nx, nst, "b", u))
        # DMA of ring position x + 3: its dY part under (x - 1, 1) ... i.e. under THIS step when st == 1 it is position (x + 4)'s dY part
        dpos, dpart = (x + 3, 1) if st == 0 else (x + 4, 0)
        dma = [th for grp in self.dma_groups(dpos, dpart) for th in grp]
        if last_of_slot_barrier:
            # start of the slot's second k-step: its fragments (the last reads of this slot) have retired -> nobody reads the slot any more;
            # ring position x + 1 has landed (two younger positions = 16 DMA instructions may stay in flight)
            nv = self.vm.need({f"dma{(x + 1) % 4}_{part}_{t}" for part in range(2) for t in range(4)})
            p.s_waitcnt(vmcnt=nv if nv is not None else 0, lgkmcnt=0)
            self.vm.wait(nv if nv is not None else 0)
            self.lg.wait(0)
            p.s_barrier()
        g = 0
        for i in range(4):
            for j in range(4):
                tags = [f"f{f}_a{i}", f"f{f}_b{j}"]
                self.wait_for(lg_tags=tags)
                p.v_mfma_f32_32x32x16_bf16(ACC(i, j), FA(f, i), FB(f, j), ACC(i, j))
                if g < len(fillers) * 2 and g % 2 == 0:
                    fillers[g // 2]()
                if g % 4 == 1 and dma:
                    dma.pop(0)()
                g += 1
        # bias gradient: the dY fragments of this k-step against ones (only where it is this workgroup's / wave's turn)
        self.uid += 1
        skip = f"L_NB{self.uid}"
        p.s_cmp("eq_u32", S_DOBIAS, 0)
        p.s_cbranch_scc1(skip)
        for u in range(4):
            for d in range(4):
                p.v_dot2c_f32_bf16(V_BG[u], FA(f, u).sub(d), V_ONES)
        p.label(skip)

    def slot(self, x):
        """one ring position: bias turn bookkeeping, two k-steps, loop exit"""
        p = self.p
        # bias turn: S_DOBIAS = db != 0 && wk == 0 && turn == 0; turn counts slots modulo ntk
        p.s_cmp("eq_u32", S_BTURN, 0)
        p.s_cselect_b32(S_DOBIAS, s(3), 0)          # s3 = (db != 0 && wk == 0)
        p.s_add_u32(S_BTURN, S_BTURN, 1)
        p.s_cmp("ge_u32", S_BTURN, S_NTK)
        p.s_cselect_b32(S_BTURN, 0, S_BTURN)
        self.kstep(x, 0, False)
        self.kstep(x, 1, True)
        p.s_sub_u32(S_LEFT, S_LEFT, 1)
        p.s_cmp("eq_u32", S_LEFT, 0)
        p.s_cbranch_scc1("L_EPI")

    # ------------------------------------------------------------------ whole kernel
    def build(self):
        p = self.p
        self.prologue()
        # s3 = bias duty of this wave
        p.s_or_b32(S_T[0], S_DB.sub(0), S_DB.sub(1))
        p.s_cmp("lg_u32", S_T[0], 0)
        p.s_cselect_b32(s(3), 1, 0)
        p.s_cmp("eq_u32", S_WK, 0)
        p.s_cselect_b32(s(3), s(3), 0)
        # seed the queue models with one scratch pass of the loop body (the loop is entered from itself at run time)
        real, self.p = self.p, Prog("scratch")
        lgq, vmq = list(self.lg.q), list(self.vm.q)
        for x in range(4):
            self.slot(x)
        self.p = real
        p.label("L_LOOP")
        for x in range(4):
            self.slot(x)
        p.s_branch("L_LOOP")
        p.label("L_EPI")
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        self.vm.wait(0)
        self.lg.wait(0)
        self.epilogue()
        p.s_endpgm()
        return self

    def epilogue(self):
        p = self.p
        T = V_T
        # dW + ((n0 + wn*128) * ldw + k0 + wk*128) * 4
        p.s_lshl_b32(S_T[0], S_WN, 7)
        p.s_add_u32(S_T[0], S_T[0], S_N0)
        p.s_mul_hi_u32(S_T[2], S_T[0], S_LDW4)
        p.s_mul_i32(S_T[1], S_T[0], S_LDW4)
        p.s_lshl_b32(S_T[3], S_WK, 7)
        p.s_add_u32(S_T[3], S_T[3], S_K0)
        p.s_lshl_b32(S_T[3], S_T[3], 2)
        p.s_add_u32(S_T[1], S_T[1], S_T[3])
        p.s_addc_u32(S_T[2], S_T[2], 0)
        p.s_add_u32(SRD_W.sub(0), S_DW.sub(0), S_T[1])
        p.s_addc_u32(S_T[2], S_DW.sub(1), S_T[2])
        p.s_and_b32(SRD_W.sub(1), S_T[2], 0xffff)
        p.s_mov_b32(SRD_W.sub(2), 0xffffffff)
        p.s_mov_b32(SRD_W.sub(3), 0x00020000)
        # lane: row 4 h of the (r & 3) + 8 (r >> 2) pattern, column c
        p.v_and_b32(T[1], 31, V_LANE)
        p.v_lshrrev_b32(T[2], 5, V_LANE)
        p.v_lshlrev_b32(T[2], 2, T[2])
        p.v_mul_lo_u32(T[2], T[2], S_LDW4)
        p.v_lshl_add_u32(V_OFF, T[1], 2, T[2])
        for i in range(4):
            for r in range(16):
                row = 32 * i + (r & 3) + 8 * (r >> 2)
                p.s_mul_i32(S_ROW, S_LDW4, row)
                for j in range(4):
                    p.buffer_atomic_add_f32(ACC(i, j).sub(r), V_OFF, SRD_W, S_ROW, 128 * j)
        # bias gradient: lanes c / c + 32 hold the sums over the two row halves of column wn*128 + 32 u + c
        self.uid += 1
        done = f"L_BD{self.uid}"
        p.s_cmp("eq_u32", s(3), 0)
        p.s_cbranch_scc1(done)
        p.s_lshl_b32(S_T[0], S_WN, 7)
        p.s_add_u32(S_T[0], S_T[0], S_N0)
        p.s_lshl_b32(S_T[0], S_T[0], 2)
        p.s_add_u32(SRD_B.sub(0), S_DB.sub(0), S_T[0])
        p.s_addc_u32(S_T[1], S_DB.sub(1), 0)
        p.s_and_b32(SRD_B.sub(1), S_T[1], 0xffff)
        p.s_mov_b32(SRD_B.sub(2), 0xffffffff)
        p.s_mov_b32(SRD_B.sub(3), 0x00020000)
        p.v_lshlrev_b32(T[3], 2, T[1])
        for u in range(4):
            p.buffer_atomic_add_f32(V_BG[u], T[3], SRD_B, 0, 128 * u)
        p.label(done)
        p.s_waitcnt(vmcnt=0)

    def asm_text(self):
        from . import nt_as_gen as G
        g = G.NtAsGen(name=self.name)
        g.p = self.p
        t = g.asm_text()
        t = t.replace(f".amdhsa_kernarg_size {G.KARG_BYTES}", f".amdhsa_kernarg_size {KARG_BYTES}").replace(
            f".kernarg_segment_size: {G.KARG_BYTES}", f".kernarg_segment_size: {KARG_BYTES}").replace(
            f".size: {G.KARG_BYTES}, .offset: 0", f".size: {KARG_BYTES}, .offset: 0")
        t = t.replace(f".amdhsa_group_segment_fixed_size {G.LDS_BYTES}", f".amdhsa_group_segment_fixed_size {LDS_BYTES}").replace(
            f".group_segment_fixed_size: {G.LDS_BYTES}", f".group_segment_fixed_size: {LDS_BYTES}")
        return t


def generate(dbg=""):
    return TnOsGen(dbg=dbg).build()
