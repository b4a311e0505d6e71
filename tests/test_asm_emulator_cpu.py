"""The generated gfx950 assembly GEMM kernels (safevla_amd/asmgen/nt_as_gen.py) executed instruction by instruction in the lane-accurate
emulator of asmgen/amdasm.py (asynchronous loads land only at the s_waitcnt that retires them; LDS-DMA targets are poisoned until then),
against a numpy restatement of the epilogues of csrc/gemm.hip (reference layers: nn.TransformerEncoderLayer linears of
architecture/models/allenact_transformer_models/allenact_dino_transformer.py:545-552).  CPU only: this is what validates a schedule
before it is ever run on a GPU; tests/test_kernels_gpu.py compares the same kernels with the HIP kernels on hardware."""
import struct

import numpy as np
import pytest

from safevla_amd.asmgen import nt_as_gen as G
from safevla_amd.asmgen.amdasm import Emu, bf16_to_f32, f32_to_bf16_rne

def _bf16(x):
    return f32_to_bf16_rne(np.asarray(x, dtype=np.float32)).astype(np.uint16)


def _drop_keep(M, N, key, thr, row_mult):
    """keep[m, n] of svla_dropout (csrc/common.h drop_bits / drop_mix) for element index (m * row_mult) * N + n"""
    e = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(row_mult * N) + np.arange(N, dtype=np.uint64)[None, :])
    pair = e >> np.uint64(1)
    x = ((pair & np.uint64(0xffffffff)) * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)
    x ^= (((pair >> np.uint64(32)) * np.uint64(0x85EBCA77)) & np.uint64(0xffffffff))
    x ^= np.uint64(key)
    x ^= x >> np.uint64(16)
    x = ((x & np.uint64(0xffffff)) * np.uint64(0xEB352D)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(13)
    x = ((x & np.uint64(0xffffff)) * np.uint64(0x6CA68B)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    r = np.where((e & np.uint64(1)) == 0, x & np.uint64(0xffff), x >> np.uint64(16))
    return r >= np.uint64(thr)


def _bits_pack(pos, N):
    """[M, N] bool -> the blocked sign-bit layout of relu_bits_word (csrc/gemm.hip): [M/32][N/64][32 rows][8 bytes]"""
    M = pos.shape[0]
    b = np.packbits(pos.reshape(M // 32, 32, N // 64, 64), axis=-1, bitorder="little")       # [M/32, 32, N/64, 8]
    return np.ascontiguousarray(b.transpose(0, 2, 1, 3)).reshape(-1)


def run_kernel(flavour, M, N, grid, order=None, seed=0, alpha=1.0, key=0x1234567, p_drop=0.1, row_mult=1, gen_kw=None, nsplit=1, flags=0, K=512):
    """grid workgroups in x (panel slots) times nsplit in y (n-ranges of N / nsplit columns: the mid-M launches)"""
    g = G.NtAsGen(name="t", **dict(G.FLAVOURS[flavour], **(gen_kw or {})))
    g.build()
    assert g.K == K
    rs = np.random.RandomState(seed)
    X = _bf16(rs.standard_normal((M, K)))
    W = _bf16(rs.standard_normal((N, K)) * 0.05)
    bias = rs.standard_normal(N).astype(np.float32) if g.bias else np.zeros(N, dtype=np.float32)
    C = np.full((M, N), 0x7fc0, dtype=np.uint16)
    thr = int(p_drop * 65536 + 0.5)
    scale = np.float32(1.0 / (1.0 - p_drop))
    bits_in_bool = rs.rand(M, N) < 0.6
    bits = _bits_pack(bits_in_bool, N) if g.bits_in else np.zeros(M * N // 8, dtype=np.uint8)
    npanels = M // 256
    q = max(N // 256, 1)
    cmask = 0 if flags & 1 else (1 << (q.bit_length() - 1)) - 1
    for wg, wy in [(x, y) for x in range(min(grid, npanels)) for y in range(nsplit)]:
        emu = Emu(g.p)
        aX, aW, aB, aC, aBits = emu.alloc(X), emu.alloc(W), emu.alloc(bias), emu.alloc(C, writable=True), emu.alloc(bits, writable=True)
        ka = bytearray(G.KARG_BYTES)

        def put(name, fmt, val):
            struct.pack_into(fmt, ka, G.KARG[name], val)
        put("A", "<Q", aX); put("lda", "<q", K); put("B", "<Q", aW); put("ldb", "<q", K); put("bias", "<Q", aB)
        put("C", "<Q", aC); put("ldc", "<q", N); put("cmask", "<i", cmask); put("N", "<i", N); put("alpha", "<f", alpha)
        put("npanels", "<i", npanels); put("grid", "<i", grid); put("bits", "<Q", aBits)
        put("key", "<I", key); put("thr", "<I", thr); put("scale", "<f", float(scale)); put("row_mult", "<i", row_mult)
        put("nr", "<i", N // nsplit); put("flags", "<i", flags)
        emu.run(ka, wg, order=order, wg_id_y=wy)
    acc = (bf16_to_f32(X).astype(np.float64) @ bf16_to_f32(W).astype(np.float64).T).astype(np.float32)
    out = bf16_to_f32(C)
    if g.bits_in:
        ref = np.where(bits_in_bool, acc * np.float32(alpha), np.float32(0))
    else:
        ref = acc + bias
        if g.drop:
            ref = np.where(_drop_keep(M, N, key, thr, row_mult), ref * scale, np.float32(0))
        if g.relu:
            ref = np.maximum(ref, 0)
        if g.gelu:
            from safevla_amd.asmgen import gelu_poly
            ref = gelu_poly.gelu_ref_np(ref)
    return out, ref, bits, g


def check(out, ref):
    assert not np.isnan(out).any(), f"{int(np.isnan(out).sum())} NaNs (a read of LDS-DMA data that had not landed, or an unwritten output)"
    err = np.abs(out - ref)
    tol = 2.0 ** -7 * np.abs(ref) + 2e-2          # one bf16 rounding + fp32 accumulation order
    assert (err <= tol).all(), f"{int((err > tol).sum())} elements off, max {err.max()}"


@pytest.mark.parametrize("order", [None, [3, 2, 1, 0]])
def test_nt_as_bias_two_panels_both_wave_orders(order):
    out, ref, _, _ = run_kernel("f0", 512, 512, 1, order=order)
    check(out, ref)


def test_nt_as_bias_three_workgroups_rotated_tiles():
    out, ref, _, _ = run_kernel("f0", 768, 1024, 3)
    check(out, ref)


def test_nt_as_relu_signbits():
    out, ref, bits, _ = run_kernel("f1", 512, 512, 1)
    check(out, ref)
    assert (bits == _bits_pack(out > 0, 512)).all()
    assert (out >= 0).all()


def test_nt_as_relu_dropout_signbits_matches_the_dropout_counter():
    out, ref, bits, _ = run_kernel("f1d", 512, 512, 2, row_mult=3)
    check(out, ref)
    assert (bits == _bits_pack(out > 0, 512)).all()
    kept = _drop_keep(512, 512, 0x1234567, int(0.1 * 65536 + 0.5), 3)
    assert (out[~kept] == 0).all() and 0.85 < kept.mean() < 0.95


def test_nt_as_bias_n_range_split_mid_m():
    """mid-M launches: grid (panel slots, nsplit), workgroup_id_y sweeps its own n-range of N / nsplit columns, no phases"""
    out, ref, _, _ = run_kernel("f0", 512, 512, 2, nsplit=2, flags=1)
    check(out, ref)
    out, ref, _, _ = run_kernel("f0", 256, 1024, 1, nsplit=4, flags=1, order=[2, 0, 3, 1])
    check(out, ref)


def test_nt_as_no_phase_persistent_two_panels():
    out, ref, _, _ = run_kernel("f0", 512, 256, 1, flags=1)      # one workgroup, two panels, every wave switches in the same step
    check(out, ref)


def test_nt_as_dropout_signbits_n_range_split_keeps_the_global_counter_and_bit_layout():
    out, ref, bits, _ = run_kernel("f1d", 256, 512, 1, nsplit=2, flags=1, row_mult=3)
    check(out, ref)
    assert (bits == _bits_pack(out > 0, 512)).all()
    kept = _drop_keep(256, 512, 0x1234567, int(0.1 * 65536 + 0.5), 3)
    assert (out[~kept] == 0).all()


def test_nt_as_signbit_mask_n_range_split():
    out, ref, _, _ = run_kernel("f3", 256, 512, 1, nsplit=2, flags=1, alpha=1.0 / 0.9)
    check(out, ref)


def test_nt_as_bias_n_384_does_not_read_past_the_bias():
    """N % 256 == 128: the bias table is loaded in 1-KiB chunks; the last chunk must be clamped by the descriptor"""
    out, ref, _, _ = run_kernel("f0", 256, 384, 1)
    check(out, ref)


def test_nt_as_k384_bias():
    """K = 384 (ViT-S width): 24 k-steps, 768-byte W rows fetched by 48 DMA lanes into the 1-KiB LDS pitch"""
    out, ref, _, _ = run_kernel("k384_f0", 512, 384, 1, K=384)
    check(out, ref)
    out, ref, _, _ = run_kernel("k384_f0", 256, 1152, 1, K=384, nsplit=3, flags=1, order=[3, 2, 1, 0])
    check(out, ref)


def test_nt_as_k384_gelu_flavour_evaluates_the_shared_polynomial():
    """bias + erf-GELU (the frozen ViT's fc1) as packed-fp32 Horner chains in the MFMA gaps: against the numpy restatement of asmgen/gelu_poly.py to one
    bf16 rounding, and against scipy's exact erf-GELU of the fp32 pre-activation"""
    from scipy.special import erf
    out, ref, _, _ = run_kernel("k384_f2", 256, 256, 1, K=384, flags=1)
    assert not np.isnan(out).any()
    err = np.abs(out - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-5).all(), float(err.max())
    pre = ref        # recompute the pre-activation for the exact form
    out2, ref2, _, _ = run_kernel("k384_f0", 256, 256, 1, K=384, flags=1)      # same seed: ref2 = acc + bias
    exact = (ref2.astype(np.float64) * 0.5 * (1 + erf(ref2.astype(np.float64) / np.sqrt(2)))).astype(np.float32)
    assert np.abs(out - exact).max() <= 2.0 ** -8 * np.abs(exact).max() + 3e-5


def test_nt_as_k384_relu_signbits():
    out, ref, bits, _ = run_kernel("k384_f1", 512, 512, 1, K=384)
    check(out, ref)
    assert (bits == _bits_pack(out > 0, 512)).all() and (out >= 0).all()


def test_nt_as_signbit_mask_alpha():
    out, ref, _, _ = run_kernel("f3", 512, 512, 1, alpha=1.0 / 0.9)
    check(out, ref)
    assert (out.view(np.uint32) != 0x80000000).all()          # masked elements are +0, as in the HIP kernels


def run_tn(M, N, Kc, chunk_rows, grid, with_bias=True, seed=0):
    from safevla_amd.asmgen import tn_os_gen as T
    g = T.generate()
    rs = np.random.RandomState(seed)
    dY = _bf16(rs.standard_normal((M, N)))
    X = _bf16(rs.standard_normal((M, Kc)))
    dW = rs.standard_normal((N, Kc)).astype(np.float32)
    db = rs.standard_normal(N).astype(np.float32)
    dW0, db0 = dW.copy(), db.copy()
    ntk, ntile = Kc // 256, (N // 256) * (Kc // 256)
    for wg in range(grid):
        emu = Emu(g.p, lds_bytes=T.LDS_BYTES)
        aY, aX, aW, aB = emu.alloc(dY), emu.alloc(X), emu.alloc(dW, writable=True), emu.alloc(db, writable=True)
        ka = bytearray(T.KARG_BYTES)

        def put(name, fmt, val):
            struct.pack_into(fmt, ka, T.KARG[name], val)
        put("dY", "<Q", aY); put("ldy", "<q", N); put("X", "<Q", aX); put("ldx", "<q", Kc); put("dW", "<Q", aW); put("ldw", "<q", Kc)
        put("db", "<Q", aB if with_bias else 0); put("M", "<i", M); put("N", "<i", N); put("K", "<i", Kc); put("chunk_rows", "<i", chunk_rows)
        put("ntile", "<i", ntile); put("ntk", "<i", ntk); put("grid", "<i", grid)
        emu.run(ka, wg)
    yf, xf = bf16_to_f32(dY).astype(np.float64), bf16_to_f32(X).astype(np.float64)
    return dW, dW0 + yf.T @ xf, db, db0 + (yf.sum(0) if with_bias else 0)


def test_tn_os_two_chunks_bias_gradient():
    # one 256 x 256 tile, two row chunks (96 + 64 rows: three slots and two), bias gradient on
    dW, rW, db, rb = run_tn(160, 256, 256, 96, 2)
    assert np.abs(dW - rW).max() < 2e-3 * np.abs(rW).max()
    assert np.abs(db - rb).max() < 2e-3 * max(1.0, np.abs(rb).max())


def test_tn_os_two_k_tiles_share_the_bias_turns():
    # 256 x 512: two tiles per chunk (k-tile 0 / 1 take the bias gradient of alternate slots), five slots (ring wrap-around)
    dW, rW, db, rb = run_tn(160, 256, 512, 160, 2)
    assert np.abs(dW - rW).max() < 2e-3 * np.abs(rW).max()
    assert np.abs(db - rb).max() < 2e-3 * max(1.0, np.abs(rb).max())


def run_nt_os(flavour, M, N, Kc, grid, order=None, seed=0):
    from safevla_amd.asmgen import nt_os_gen as O
    g = O.generate(flavour)
    rs = np.random.RandomState(seed)
    X = _bf16(rs.standard_normal((M, Kc)))
    W = _bf16(rs.standard_normal((N, Kc)) * 0.05)
    bias = rs.standard_normal(N).astype(np.float32)
    R = _bf16(rs.standard_normal((M, N)))
    C = np.full((M, N), 0x7fc0, dtype=np.uint16)
    ntn, ntiles = N // 256, (M // 256) * (N // 256)
    grid = min(grid, ntiles)
    for wg in range(grid):
        emu = Emu(g.p)
        aX, aW, aB, aR, aC = emu.alloc(X), emu.alloc(W), emu.alloc(bias), emu.alloc(R), emu.alloc(C, writable=True)
        ka = bytearray(O.KARG_BYTES)

        def put(name, fmt, val):
            struct.pack_into(fmt, ka, O.KARG[name], val)
        put("A", "<Q", aX); put("lda", "<q", Kc); put("B", "<Q", aW); put("ldb", "<q", Kc); put("bias", "<Q", aB); put("res", "<Q", aR); put("ldr", "<q", N)
        put("C", "<Q", aC); put("ldc", "<q", N); put("M", "<i", M); put("N", "<i", N); put("K", "<i", Kc); put("ntn", "<i", ntn)
        put("ntiles", "<i", ntiles); put("grid", "<i", grid)
        emu.run(ka, wg, order=order)
    ref = (bf16_to_f32(X).astype(np.float64) @ bf16_to_f32(W).astype(np.float64).T).astype(np.float32)
    if g.bias:
        ref = ref + bias
    if g.res:
        ref = ref + bf16_to_f32(R)
    return bf16_to_f32(C), ref


@pytest.mark.parametrize("flavour,M,N,Kc,grid,order", [
    ("p", 256, 256, 384, 1, None),               # one tile, one pass of the pair loop
    ("br", 512, 512, 384, 2, [3, 2, 1, 0]),      # two tiles per workgroup (deferred stores under the next tile's K-tiles), other wave order
    ("r", 768, 256, 640, 3, None),               # three workgroups, two passes of the pair loop, residual prefetch behind the draining slabs
    ("b", 256, 512, 512, 2, None),
    ("r", 512, 1024, 384, 3, None),              # 8 tiles on 3 workgroups, 4 n-tiles per row tile: the (row tile, n-tile) cursors carry (grid % ntn = 3)
])
def test_nt_os_output_stationary_kernel(flavour, M, N, Kc, grid, order):
    """svla_nt_os_* (asmgen/nt_os_gen.py): ring of released-one-by-one units, counted waits, exposed pack + deferred stores, DOT-result hazard"""
    out, ref = run_nt_os(flavour, M, N, Kc, grid, order=order)
    check(out, ref)


def test_emulator_rejects_dot_result_read_by_another_valu_too_early():
    """The hazard the MI355X taught the builder: a DOT result is not forwarded to other VALU instructions for 3 wait states and
    the hardware does not interlock.  The emulator must refuse such a stream (and accept it with the distance kept), or the next kernel will ship the bug."""
    from safevla_amd.asmgen.amdasm import EmuError, Prog, v

    def prog(gap):
        p = Prog("t")
        p.v_mov_b32(v(1), 0x3f803f80)
        p.v_mov_b32(v(2), 0x3f803f80)
        p.v_mov_b32(v(3), 0)
        p.v_dot2c_f32_bf16(v(3), v(1), v(2))
        for _ in range(gap):
            p.v_mov_b32(v(5), 0)
        p.v_add_f32(v(4), v(3), v(3))
        p.s_endpgm()
        return p
    with pytest.raises(EmuError, match="DOT"):
        Emu(prog(2), nwaves=1).run(bytes(8), 0)
    waves = Emu(prog(3), nwaves=1).run(bytes(8), 0)
    assert (waves[0].V[4].view(np.float32) == 4.0).all()
