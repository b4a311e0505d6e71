"""Hardware probes of the assembly builder's assumptions (tools/asm_probe.py): what the launch puts into s2 / v0, the raw buffer descriptor path,
the saddr form of global_load_lds, MFMA operands in AGPRs, buffer loads into AGPRs.  Built only with SVLA_ASM_DEBUG_VARIANTS=1."""
from .amdasm import M0, Prog, a, s, v
from . import nt_as_gen as G


class ProbeGen:
    name = "svla_probe"

    def __init__(self):
        self.p = Prog(self.name)

    def build(self):
        p = self.p
        # kernarg: out (8) src (8) n (4)
        p.s_load(s(4, 8), s(0, 2), 0)
        p.v_and_b32(v(1), 63, v(0))
        p.v_lshrrev_b32(v(2), 6, v(0))
        p.v_readfirstlane_b32(s(36), v(2))
        p.s_waitcnt(lgkmcnt=0)
        # SRD over out
        p.s_mov_b32(s(40), s(4))
        p.s_and_b32(s(41), s(5), 0xffff)
        p.s_mov_b32(s(42), 0xffffffff)
        p.s_mov_b32(s(43), 0x00020000)
        # SRD over src
        p.s_mov_b32(s(44), s(6))
        p.s_and_b32(s(45), s(7), 0xffff)
        p.s_mov_b32(s(46), 0xffffffff)
        p.s_mov_b32(s(47), 0x00020000)
        # record -1 (last 1 KiB of the 4th 64-KiB block): [s2, v0, s3, s4.lo] by true lane id (v_mbcnt), bounded descriptor: cannot fault
        p.s_mov_b32(s(60), s(4))
        p.s_and_b32(s(61), s(5), 0xffff)
        p.s_mov_b32(s(62), 262144)
        p.s_mov_b32(s(63), 0x00020000)
        p.v_mbcnt_lane_id(v(24))
        p.v_lshlrev_b32(v(25), 4, v(24))
        p.v_mov_b32(v(26), s(2))
        p.v_mov_b32(v(27), v(0))
        p.v_mov_b32(v(28), s(3))
        p.v_mov_b32(v(29), s(36))
        p.s_mov_b32(s(64), 261120)
        p.buffer_store(v(26, 4), v(25), s(60, 4), s(64))
        p.s_waitcnt(vmcnt=0)
        p.s_cmp("lg_u32", s(8), 0)
        p.s_cbranch_scc0("L_GO")
        p.s_endpgm()
        p.label("L_GO")
        # record 0: [s2, v0, wave, 0x1234] at out[(s2 * 256 + tid) * 16]
        p.s_lshl_b32(s(50), s(2), 12)          # workgroup * 256 threads * 16 B
        p.v_lshlrev_b32(v(3), 4, v(0))
        p.v_mov_b32(v(4), s(2))
        p.v_mov_b32(v(5), v(0))
        p.v_mov_b32(v(6), s(36))
        p.v_mov_b32(v(7), 0x1234)
        p.buffer_store(v(4, 4), v(3), s(40, 4), s(50))
        # record 1 (at + 64 KiB): buffer_load_dwordx4 into AGPRs from src[tid * 16], moved to VGPRs, stored
        p.buffer_load(a(8, 4), v(3), s(44, 4), 0)
        p.s_waitcnt(vmcnt=0)
        for i in range(4):
            p.v_accvgpr_read_b32(v(8 + i), a(8 + i))
        p.s_add_u32(s(51), s(50), 65536)
        p.buffer_store(v(8, 4), v(3), s(40, 4), s(51))
        # record 2 (at + 128 KiB): global_load_lds saddr form: wave w fetches src[w * 1024 + (lane ^ 5) * 16] to LDS[70000 + 16 + w * 1024 + lane * 16]
        p.v_lshlrev_b32(v(12), 4, v(1))
        p.v_xor_b32(v(13), 0x50, v(12))
        p.s_lshl_b32(s(52), s(36), 10)
        p.s_add_u32(s(54), s(6), s(52))
        p.s_addc_u32(s(55), s(7), 0)
        p.s_add_u32(M0, s(52), 70016)
        p.s_nop(0)
        p.global_load_lds_x4(v(13), s(54, 2))
        p.s_waitcnt(vmcnt=0)
        p.s_barrier()
        p.v_add_u32(v(14), 70016, v(3))
        p.ds_read(v(16, 4), v(14))
        p.s_waitcnt(lgkmcnt=0)
        p.s_add_u32(s(51), s(50), 131072)
        p.buffer_store(v(16, 4), v(3), s(40, 4), s(51))
        # record 3 (at + 192 KiB): MFMA with the B operand in AGPRs: A = src fragment (VGPR), B = same data in AGPRs, C = 0; store acc[0:3]
        p.buffer_load(v(20, 4), v(3), s(44, 4), 0)
        p.s_waitcnt(vmcnt=0)
        p.v_mfma_f32_32x32x16_bf16(v(32, 16), v(20, 4), a(8, 4), 0)
        p.s_nop(7)
        p.s_nop(7)
        p.s_nop(7)
        p.s_add_u32(s(51), s(50), 196608)
        p.buffer_store(v(32, 4), v(3), s(40, 4), s(51))
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        p.s_endpgm()
        return self

    def asm_text(self):
        g = G.NtAsGen(name=self.name)
        g.p = self.p
        return g.asm_text().replace(f".amdhsa_kernarg_size {G.KARG_BYTES}", ".amdhsa_kernarg_size 32").replace(
            f".kernarg_segment_size: {G.KARG_BYTES}", ".kernarg_segment_size: 32").replace(f".size: {G.KARG_BYTES}, .offset: 0", ".size: 32, .offset: 0")
