"""A small gfx950 assembly builder with a lane-accurate CPU emulator of the instructions it can emit.

Why it exists: the A-stationary GEMM kernels (nt_as_gen.py) are hand-scheduled instruction streams -- one wave per SIMD, 256 AGPRs of
stationary operand, MFMA / LDS / LDS-DMA / epilogue VALU interleaved by a generator, every ``s_waitcnt`` counted by the generator.  hipcc
cannot produce that schedule ("what the GEMM ablations point to next"), and this container has no GPU, so every builder method
does two things: it appends the instruction's TEXT (assembled by clang for gfx950) and a Python closure with the instruction's
SEMANTICS.  ``Emu`` runs the closures for the four waves of a workgroup over numpy memory.

The emulator is adversarial about asynchrony -- the part a functional model usually gets wrong and the part a hand-written stream gets wrong:
  * a vector-memory or LDS load's destination registers are "pending" from issue until an ``s_waitcnt`` retires the operation (in order,
    oldest first, exactly as many as the count demands and no more); reading or overwriting a pending register raises;
  * LDS-DMA data lands only when the ISSUING wave retires it, and the target bytes are poisoned (bf16 NaN) at issue, so a read that is
    not ordered by  vmcnt -> s_barrier  sees NaNs, and so does a DMA issued over bytes another wave has yet to read (WAR) when that wave
    is scheduled later -- run both wave orders;
  * global stores become visible when retired; lgkmcnt / vmcnt saturate at their hardware widths (4 / 6 bits).
What it does not model: issue timing, MFMA/VALU wait-state hazards (the generator's own rules keep consumers far from producers), caches.
Test infrastructure + build-time generator only: nothing on the product path imports the emulator.
"""
import struct
from dataclasses import dataclass

import numpy as np

LANES = np.arange(64, dtype=np.int64)
BF16_NAN = 0x7FC0


@dataclass(frozen=True)
class Reg:
    kind: str   # 'v', 'a', 's'
    idx: int
    n: int = 1

    def __str__(self):
        if self.n == 1:
            return f"{self.kind}{self.idx}"
        return f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def __post_init__(self):
        # gfx90a+: vector register tuples must start at an even register
        assert not (self.kind in ("v", "a") and self.n > 1 and self.idx % 2), f"misaligned register tuple {self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def sub(self, i, n=1):
        assert 0 <= i and i + n <= self.n, (self, i, n)
        return Reg(self.kind, self.idx + i, n)


def v(i, n=1):
    return Reg("v", i, n)


def a(i, n=1):
    return Reg("a", i, n)


def s(i, n=1):
    return Reg("s", i, n)


VCC = Reg("vcc", 0, 2)
EXEC = Reg("exec", 0, 2)
M0 = Reg("m0", 0, 1)


def _txt(x):
    if isinstance(x, Reg):
        if x.kind in ("vcc", "exec", "m0"):
            return x.kind
        return str(x)
    if isinstance(x, float):
        return f"0x{struct.unpack('<I', struct.pack('<f', x))[0]:08x}"
    if isinstance(x, int):
        if -16 <= x <= 64:
            return str(x)
        return f"0x{x & 0xffffffff:08x}"
    raise TypeError(x)


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


class EmuError(RuntimeError):
    pass


class Wave:
    def __init__(self, wid, emu):
        self.wid = wid
        self.emu = emu
        self.V = np.zeros((256, 64), dtype=np.uint32)
        self.A = np.zeros((256, 64), dtype=np.uint32)
        self.S = np.zeros(128, dtype=np.uint32)
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.m0 = 0
        self.scc = 0
        self.pc = 0
        self.pend = {}       # (kind, idx) -> number of outstanding loads writing it
        self.vmq = []        # outstanding vector-memory operations (callables run at retirement), oldest first
        self.lgq = []
        self.done = False
        self.at_barrier = False
        self.counts = {}
        self.nstep = 0       # instructions (wait states) executed
        self.dotw = {}       # VGPR index -> nstep of the DOT instruction that wrote it (its result is not forwarded to other VALU operations)
        self.in_dot = False
        # random garbage in the register files: uninitialised reads must not pass by luck
        rs = np.random.RandomState(1234 + wid)
        self.V[:] = rs.randint(0, 2**32, size=self.V.shape, dtype=np.uint64).astype(np.uint32)
        self.A[:] = rs.randint(0, 2**32, size=self.A.shape, dtype=np.uint64).astype(np.uint32)

    # ---- register access
    def _chk(self, r, what):
        for i in range(r.n):
            if self.pend.get((r.kind, r.idx + i), 0):
                raise EmuError(f"wave {self.wid} pc {self.pc}: {what} of {r.kind}{r.idx + i} while a load into it is outstanding: {self.emu.prog.ops[self.pc].text}")

    def lanes(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], dtype=bool)

    def rd(self, x, i=0):
        """one dword of operand x as uint32[64]"""
        if isinstance(x, Reg):
            if x.kind == "v":
                self._chk(x.sub(i), "read")
                # gfx940-family hazard the hardware does not interlock (found on an MI355X, LLVM: DotWriteDifferentVALURead = 3 wait states):
                # the result of a DOT instruction read by anything but another DOT less than 4 instructions later is the OLD value
                t = self.dotw.get(x.idx + i)
                if t is not None and not self.in_dot and self.nstep - t < 4:
                    raise EmuError(f"wave {self.wid} pc {self.pc}: v{x.idx + i} written by a DOT instruction {self.nstep - t} instruction(s) ago is read by "
                                   f"a non-DOT instruction (3 wait states required): {self.emu.prog.ops[self.pc].text}")
                return self.V[x.idx + i].copy()
            if x.kind == "a":
                self._chk(x.sub(i), "read")
                return self.A[x.idx + i].copy()
            if x.kind == "s":
                return np.full(64, self.S[x.idx + i], dtype=np.uint32)
            if x.kind == "vcc":
                return np.full(64, (self.vcc >> (32 * i)) & 0xffffffff, dtype=np.uint32)
            if x.kind == "m0":
                return np.full(64, self.m0, dtype=np.uint32)
            raise EmuError(x)
        if isinstance(x, float):
            return np.full(64, f2u(x), dtype=np.uint32)
        return np.full(64, x & 0xffffffff, dtype=np.uint32)

    def wr(self, r, val, i=0, masked=True):
        val = np.asarray(val).astype(np.uint32)
        self._chk(r.sub(i), "write")
        arr = self.V if r.kind == "v" else self.A
        if masked and self.exec != (1 << 64) - 1:
            m = self.lanes()
            arr[r.idx + i][m] = val[m]
        else:
            arr[r.idx + i] = val

    def srd(self, x, i=0):
        if isinstance(x, Reg):
            if x.kind == "s":
                return int(self.S[x.idx + i])
            if x.kind == "m0":
                return self.m0
            if x.kind == "vcc":
                return (self.vcc >> (32 * i)) & 0xffffffff
            if x.kind == "exec":
                return (self.exec >> (32 * i)) & 0xffffffff
            raise EmuError(f"scalar read of {x}")
        if isinstance(x, float):
            return f2u(x)
        return x & 0xffffffff

    def swr(self, r, val, i=0):
        val &= 0xffffffff
        if r.kind == "s":
            self.S[r.idx + i] = val
        elif r.kind == "m0":
            self.m0 = val
        elif r.kind == "vcc":
            self.vcc = (self.vcc & ~(0xffffffff << (32 * i))) | (val << (32 * i))
        elif r.kind == "exec":
            self.exec = (self.exec & ~(0xffffffff << (32 * i))) | (val << (32 * i))
        else:
            raise EmuError(f"scalar write of {r}")

    def mark(self, r):
        for i in range(r.n):
            k = (r.kind, r.idx + i)
            self.pend[k] = self.pend.get(k, 0) + 1

    def unmark(self, r):
        for i in range(r.n):
            k = (r.kind, r.idx + i)
            self.pend[k] -= 1

    def retire(self, q, n):
        while len(q) > n:
            q.pop(0)()


@dataclass
class Op:
    text: str
    fn: object
    kind: str


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_rne(f):
    u = np.asarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint32)


class Prog:
    """Instruction stream builder.  Every method appends text + semantics."""

    def __init__(self, name):
        self.name = name
        self.ops = []
        self.labels = {}

    # ------------------------------------------------------------------ plumbing
    def _add(self, text, fn, kind):
        self.ops.append(Op(text, fn, kind))

    def label(self, name):
        assert name not in self.labels, name
        self.labels[name] = len(self.ops)
        self._add(f"{name}:", None, "label")

    def comment(self, text):
        self._add(f"// {text}", None, "label")

    def text(self):
        out = []
        for o in self.ops:
            out.append(o.text if o.kind == "label" else "\t" + o.text)
        return "\n".join(out) + "\n"

    def count(self, kind=None):
        return sum(1 for o in self.ops if o.kind not in ("label",) and (kind is None or o.kind == kind))

    # ------------------------------------------------------------------ SALU
    def _salu(self, text, fn):
        self._add(text, fn, "salu")

    def s_mov_b32(self, d, x):
        self._salu(f"s_mov_b32 {_txt(d)}, {_txt(x)}", lambda w: w.swr(d, w.srd(x)))

    def s_mov_b64(self, d, x):
        def fn(w):
            if isinstance(x, Reg):
                lo, hi = w.srd(x, 0), w.srd(x, 1)
            else:
                val = x & 0xffffffffffffffff if x >= 0 else (x + (1 << 64))
                lo, hi = val & 0xffffffff, val >> 32
            w.swr(d, lo, 0)
            w.swr(d, hi, 1)
        self._salu(f"s_mov_b64 {_txt(d)}, {_txt(x) if isinstance(x, Reg) else x}", fn)

    def _s2(self, name, d, x, y, f, scc=None):
        def fn(w):
            r = f(w.srd(x), w.srd(y), w)
            if scc is not None:
                w.scc = int(scc(w.srd(x), w.srd(y), r, w))
            w.swr(d, r)
        self._salu(f"{name} {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def s_add_u32(self, d, x, y):
        self._s2("s_add_u32", d, x, y, lambda p, q, w: p + q, lambda p, q, r, w: (p + q) >> 32)

    def s_addc_u32(self, d, x, y):
        self._s2("s_addc_u32", d, x, y, lambda p, q, w: p + q + w.scc, lambda p, q, r, w: (p + q + w.scc) >> 32)

    def s_sub_u32(self, d, x, y):
        self._s2("s_sub_u32", d, x, y, lambda p, q, w: p - q, lambda p, q, r, w: q > p)

    def s_add_i32(self, d, x, y):
        self._s2("s_add_i32", d, x, y, lambda p, q, w: p + q, lambda p, q, r, w: 0)      # SCC = signed overflow: never relied on

    def s_sub_i32(self, d, x, y):
        self._s2("s_sub_i32", d, x, y, lambda p, q, w: p - q, lambda p, q, r, w: 0)

    def s_mul_i32(self, d, x, y):
        self._s2("s_mul_i32", d, x, y, lambda p, q, w: p * q)

    def s_mul_hi_u32(self, d, x, y):
        self._s2("s_mul_hi_u32", d, x, y, lambda p, q, w: (p * q) >> 32)

    def s_lshl_b32(self, d, x, y):
        self._s2("s_lshl_b32", d, x, y, lambda p, q, w: p << (q & 31), lambda p, q, r, w: (r & 0xffffffff) != 0)

    def s_lshr_b32(self, d, x, y):
        self._s2("s_lshr_b32", d, x, y, lambda p, q, w: p >> (q & 31), lambda p, q, r, w: (r & 0xffffffff) != 0)

    def s_and_b32(self, d, x, y):
        self._s2("s_and_b32", d, x, y, lambda p, q, w: p & q, lambda p, q, r, w: (r & 0xffffffff) != 0)

    def s_or_b32(self, d, x, y):
        self._s2("s_or_b32", d, x, y, lambda p, q, w: p | q, lambda p, q, r, w: (r & 0xffffffff) != 0)

    def s_xor_b32(self, d, x, y):
        self._s2("s_xor_b32", d, x, y, lambda p, q, w: p ^ q, lambda p, q, r, w: (r & 0xffffffff) != 0)

    def s_and_b64(self, d, x, y):
        def fn(w):
            def val(o):
                if isinstance(o, Reg):
                    return w.srd(o, 0) | (w.srd(o, 1) << 32)
                return o & 0xffffffffffffffff if o >= 0 else o + (1 << 64)
            r = val(x) & val(y)
            w.scc = int(r != 0)
            w.swr(d, r & 0xffffffff, 0)
            w.swr(d, r >> 32, 1)
        self._salu(f"s_and_b64 {_txt(d)}, {_txt(x) if isinstance(x, Reg) else x}, {_txt(y) if isinstance(y, Reg) else y}", fn)

    def s_min_u32(self, d, x, y):
        self._s2("s_min_u32", d, x, y, lambda p, q, w: min(p, q), lambda p, q, r, w: p < q)

    def s_cmp(self, op, x, y):
        f = {"lt_u32": lambda p, q: p < q, "ge_u32": lambda p, q: p >= q, "eq_u32": lambda p, q: p == q, "lg_u32": lambda p, q: p != q,
             "gt_u32": lambda p, q: p > q, "le_u32": lambda p, q: p <= q}[op]

        def fn(w):
            w.scc = int(f(w.srd(x), w.srd(y)))
        self._salu(f"s_cmp_{op} {_txt(x)}, {_txt(y)}", fn)

    def s_cselect_b32(self, d, x, y):
        self._salu(f"s_cselect_b32 {_txt(d)}, {_txt(x)}, {_txt(y)}", lambda w: w.swr(d, w.srd(x) if w.scc else w.srd(y)))

    def s_cselect_b64(self, d, x, y):
        def fn(w):
            src = x if w.scc else y
            if isinstance(src, Reg):
                lo, hi = w.srd(src, 0), w.srd(src, 1)
            else:
                val = src & 0xffffffffffffffff if src >= 0 else (src + (1 << 64))
                lo, hi = val & 0xffffffff, val >> 32
            w.swr(d, lo, 0)
            w.swr(d, hi, 1)
        self._salu(f"s_cselect_b64 {_txt(d)}, {_txt(x) if isinstance(x, Reg) else x}, {_txt(y) if isinstance(y, Reg) else y}", fn)

    def s_load(self, d, base, off):
        """s_load_dword{,x2,x4,x8,x16} d, base, off (kernarg loads; retired by lgkmcnt(0) only: SMEM returns out of order)"""
        suffix = {1: "dword", 2: "dwordx2", 4: "dwordx4", 8: "dwordx8", 16: "dwordx16"}[d.n]

        def fn(w):
            addr = w.srd(base, 0) | (w.srd(base, 1) << 32)
            data = w.emu.mem_read(addr + off, 4 * d.n).view(np.uint32)
            vals = [int(x) for x in data]

            def land():
                for i, x in enumerate(vals):
                    w.swr(d, x, i)
            w.lgq.append(land)
            w.smem_out = getattr(w, "smem_out", 0) + 1
        self._add(f"s_load_{suffix} {_txt(d)}, {_txt(base)}, 0x{off:x}", fn, "smem")

    def s_waitcnt(self, vmcnt=None, lgkmcnt=None):
        parts = []
        if vmcnt is not None:
            assert 0 <= vmcnt <= 63, vmcnt
            parts.append(f"vmcnt({vmcnt})")
        if lgkmcnt is not None:
            assert 0 <= lgkmcnt <= 15, lgkmcnt
            parts.append(f"lgkmcnt({lgkmcnt})")
        assert parts

        def fn(w):
            if vmcnt is not None:
                w.retire(w.vmq, vmcnt)
            if lgkmcnt is not None:
                if getattr(w, "smem_out", 0) and lgkmcnt != 0:
                    raise EmuError("counted lgkmcnt with scalar loads outstanding (SMEM returns out of order)")
                w.retire(w.lgq, lgkmcnt)
                if lgkmcnt == 0:
                    w.smem_out = 0
        self._add("s_waitcnt " + " ".join(parts), fn, "wait")

    def s_barrier(self):
        def fn(w):
            w.at_barrier = True
        self._add("s_barrier", fn, "barrier")

    def s_nop(self, n=0):
        def fn(w):
            w.nstep += n          # n + 1 wait states
        self._salu(f"s_nop {n}", fn)

    def s_setprio(self, n):
        self._salu(f"s_setprio {n}", lambda w: None)

    def s_endpgm(self):
        def fn(w):
            if w.vmq or w.lgq:
                raise EmuError(f"wave {w.wid}: s_endpgm with {len(w.vmq)} VM / {len(w.lgq)} LGKM operations outstanding")
            w.done = True
        self._add("s_endpgm", fn, "end")

    def s_branch(self, label):
        def fn(w):
            w.pc = w.emu.prog.labels[label] - 1
        self._add(f"s_branch {label}", fn, "branch")

    def s_cbranch_scc1(self, label):
        def fn(w):
            if w.scc:
                w.pc = w.emu.prog.labels[label] - 1
        self._add(f"s_cbranch_scc1 {label}", fn, "branch")

    def s_cbranch_scc0(self, label):
        def fn(w):
            if not w.scc:
                w.pc = w.emu.prog.labels[label] - 1
        self._add(f"s_cbranch_scc0 {label}", fn, "branch")

    def s_memtime(self, d):
        """64-bit shader clock through the scalar memory path: returns on lgkmcnt, out of order -- follow it with lgkmcnt(0)"""
        def fn(w):
            w.emu.clock = getattr(w.emu, "clock", 0) + 1000

            def land(t=w.emu.clock):
                w.swr(d, t, 0)
                w.swr(d, 0, 1)
            w.lgq.append(land)
            w.smem_out = getattr(w, "smem_out", 0) + 1
        self._add(f"s_memtime {_txt(d)}", fn, "smem")

    # ------------------------------------------------------------------ VALU
    def _valu(self, text, fn):
        self._add(text, fn, "valu")

    def _v2(self, name, d, x, y, f):
        self._valu(f"{name} {_txt(d)}, {_txt(x)}, {_txt(y)}", lambda w: w.wr(d, f(w.rd(x).astype(np.uint64), w.rd(y).astype(np.uint64)) & 0xffffffff))

    def _v3(self, name, d, x, y, z, f):
        self._valu(f"{name} {_txt(d)}, {_txt(x)}, {_txt(y)}, {_txt(z)}",
                   lambda w: w.wr(d, f(w.rd(x).astype(np.uint64), w.rd(y).astype(np.uint64), w.rd(z).astype(np.uint64)) & 0xffffffff))

    def v_mov_b32(self, d, x):
        self._valu(f"v_mov_b32 {_txt(d)}, {_txt(x)}", lambda w: w.wr(d, w.rd(x)))

    def v_swap_b32(self, x, y):
        def fn(w):
            a_, b_ = w.rd(x), w.rd(y)
            w.wr(x, b_)
            w.wr(y, a_)
        self._valu(f"v_swap_b32 {_txt(x)}, {_txt(y)}", fn)

    def v_accvgpr_write_b32(self, d, x):
        self._valu(f"v_accvgpr_write_b32 {_txt(d)}, {_txt(x)}", lambda w: w.wr(d, w.rd(x)))

    def v_accvgpr_read_b32(self, d, x):
        self._valu(f"v_accvgpr_read_b32 {_txt(d)}, {_txt(x)}", lambda w: w.wr(d, w.rd(x)))

    def v_add_u32(self, d, x, y):
        self._v2("v_add_u32", d, x, y, lambda p, q: p + q)

    def v_sub_u32(self, d, x, y):
        self._v2("v_sub_u32", d, x, y, lambda p, q: p - q + (1 << 32))

    def v_lshlrev_b32(self, d, sh, x):
        self._v2("v_lshlrev_b32", d, sh, x, lambda p, q: q << (p & 31))

    def v_lshrrev_b32(self, d, sh, x):
        self._v2("v_lshrrev_b32", d, sh, x, lambda p, q: q >> (p & 31))

    def v_and_b32(self, d, x, y):
        self._v2("v_and_b32", d, x, y, lambda p, q: p & q)

    def v_or_b32(self, d, x, y):
        self._v2("v_or_b32", d, x, y, lambda p, q: p | q)

    def v_xor_b32(self, d, x, y):
        self._v2("v_xor_b32", d, x, y, lambda p, q: p ^ q)

    def v_mul_lo_u32(self, d, x, y):
        self._v2("v_mul_lo_u32", d, x, y, lambda p, q: p * q)

    def v_mul_u32_u24(self, d, x, y):
        self._v2("v_mul_u32_u24", d, x, y, lambda p, q: (p & 0xffffff) * (q & 0xffffff))

    def v_mad_u32_u24(self, d, x, y, z):
        self._v3("v_mad_u32_u24", d, x, y, z, lambda p, q, r: (p & 0xffffff) * (q & 0xffffff) + r)

    def v_lshl_add_u32(self, d, x, sh, z):
        self._v3("v_lshl_add_u32", d, x, sh, z, lambda p, q, r: (p << (q & 31)) + r)

    def v_add_lshl_u32(self, d, x, y, sh):
        self._v3("v_add_lshl_u32", d, x, y, sh, lambda p, q, r: ((p + q) & 0xffffffff) << (r & 31))

    def v_lshl_or_b32(self, d, x, sh, z):
        self._v3("v_lshl_or_b32", d, x, sh, z, lambda p, q, r: (p << (q & 31)) | r)

    def v_and_or_b32(self, d, x, y, z):
        self._v3("v_and_or_b32", d, x, y, z, lambda p, q, r: (p & q) | r)

    def v_add3_u32(self, d, x, y, z):
        self._v3("v_add3_u32", d, x, y, z, lambda p, q, r: p + q + r)

    def v_bfe_u32(self, d, x, off, width):
        self._v3("v_bfe_u32", d, x, off, width, lambda p, q, r: (p >> (q & 31)) & ((1 << (r & 31)) - 1))

    def v_mbcnt_lane_id(self, d):
        """lane index 0..63 (v_mbcnt_lo + v_mbcnt_hi over an all-ones mask)"""
        self._valu(f"v_mbcnt_lo_u32_b32 {_txt(d)}, -1, 0", lambda w: w.wr(d, np.minimum(LANES, 32).astype(np.uint32)))
        self._valu(f"v_mbcnt_hi_u32_b32 {_txt(d)}, -1, {_txt(d)}", lambda w: w.wr(d, LANES.astype(np.uint32)))

    def v_readfirstlane_b32(self, d, x):
        # gfx940-family hazard (measured: tools/asm_probe.py returned a stale register without it): a VALU write of x needs a wait state before
        # v_readfirstlane reads it, and the SGPR it writes needs 5 before a vector-memory instruction uses it -- pad both sides here, always
        self.s_nop(1)
        self._v_readfirstlane_raw(d, x)
        self.s_nop(4)

    def _v_readfirstlane_raw(self, d, x):
        def fn(w):
            lanes = np.nonzero(w.lanes())[0]
            w.swr(d, int(w.rd(x)[lanes[0] if len(lanes) else 0]))
        self._valu(f"v_readfirstlane_b32 {_txt(d)}, {_txt(x)}", fn)

    def _vf2(self, name, d, x, y, f):
        def fn(w):
            p, q = w.rd(x).view(np.float32), w.rd(y).view(np.float32)
            with np.errstate(all="ignore"):
                w.wr(d, f(p, q).astype(np.float32).view(np.uint32))
        self._valu(f"{name} {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def v_mul_f32(self, d, x, y):
        self._vf2("v_mul_f32", d, x, y, lambda p, q: p * q)

    def v_add_f32(self, d, x, y):
        self._vf2("v_add_f32", d, x, y, lambda p, q: p + q)

    def v_max_f32(self, d, x, y):
        self._vf2("v_max_f32", d, x, y, np.maximum)

    def v_med3_f32(self, d, x, y, z, neg=(0, 0, 0)):
        """median of three fp32 (a clamp when two operands are the bounds); neg: per-source negation modifiers (the same SGPR may serve as -c and +c:
        one constant-bus read)"""
        def tx(o, n):
            return ("-" if n else "") + (_txt(o) if not isinstance(o, float) else repr(o))

        def fn(w):
            vals = []
            for o, n in zip((x, y, z), neg):
                f = w.rd(o).view(np.float32).astype(np.float32)
                vals.append(-f if n else f)
            with np.errstate(all="ignore"):
                st = np.sort(np.stack(vals), axis=0)[1]
            w.wr(d, st.astype(np.float32).view(np.uint32))
        for o in (x, y, z):
            assert not isinstance(o, float) or o in (0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0, 0.0), "VOP3 on gfx9 takes inline constants only"
        self._valu(f"v_med3_f32 {_txt(d)}, {tx(x, neg[0])}, {tx(y, neg[1])}, {tx(z, neg[2])}", fn)

    def _pk_f32(self, name, d, srcs, sel, f):
        """packed fp32 (two fp32 operations per lane on 64-bit register pairs).  sel[i] = (dword of source i that feeds the LOW result, dword that feeds
        the HIGH result): (0, 1) = the pair as it is, (0, 0) / (1, 1) = its low / high dword broadcast (op_sel / op_sel_hi) -- how one SGPR pair carries
        two constants.  gfx9: at most one SGPR (pair) per instruction, no literals."""
        assert d.n == 2 and all(isinstance(o, Reg) and o.n == 2 for o in srcs), "packed fp32 operands are 64-bit register pairs"
        assert sum(1 for o in srcs if o.kind == "s") <= 1 or len({(o.kind, o.idx) for o in srcs if o.kind == "s"}) == 1, "constant bus: one SGPR pair"
        sel = list(sel) + [(0, 1)] * (len(srcs) - len(sel))
        op_sel, op_sel_hi = [lo for lo, _ in sel], [hi for _, hi in sel]
        mods = ""
        if any(op_sel):
            mods += " op_sel:[" + ",".join(str(b) for b in op_sel) + "]"
        if not all(op_sel_hi):
            mods += " op_sel_hi:[" + ",".join(str(b) for b in op_sel_hi) + "]"

        def fn(w):
            out = []
            for half in (0, 1):
                with np.errstate(all="ignore"):
                    args = [w.rd(o, sl[half]).view(np.float32).astype(np.float64) for o, sl in zip(srcs, sel)]
                    out.append(f(*args).astype(np.float32).view(np.uint32))      # one rounding (the fp64 intermediate holds the product exactly)
            w.wr(d, out[0], 0)
            w.wr(d, out[1], 1)
        self._valu(f"{name} {_txt(d)}, " + ", ".join(_txt(o) for o in srcs) + mods, fn)

    def v_pk_mul_f32(self, d, x, y, sel=()):
        self._pk_f32("v_pk_mul_f32", d, (x, y), sel, lambda p, q: p * q)

    def v_pk_add_f32(self, d, x, y, sel=()):
        self._pk_f32("v_pk_add_f32", d, (x, y), sel, lambda p, q: p + q)

    def v_pk_fma_f32(self, d, x, y, z, sel=()):
        self._pk_f32("v_pk_fma_f32", d, (x, y, z), sel, lambda p, q, r: p * q + r)

    def v_cvt_pk_bf16_f32(self, d, x, y):
        def fn(w):
            lo = f32_to_bf16_rne(w.rd(x).view(np.float32))
            hi = f32_to_bf16_rne(w.rd(y).view(np.float32))
            w.wr(d, lo | (hi << 16))
        self._valu(f"v_cvt_pk_bf16_f32 {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def v_pk_max_i16(self, d, x, y):
        def fn(w):
            p, q = w.rd(x), w.rd(y)
            out = np.zeros(64, dtype=np.uint32)
            for sh in (0, 16):
                pa = ((p >> sh) & 0xffff).astype(np.uint16).view(np.int16)
                qa = ((q >> sh) & 0xffff).astype(np.uint16).view(np.int16)
                out |= np.maximum(pa, qa).view(np.uint16).astype(np.uint32) << sh
            w.wr(d, out)
        self._valu(f"v_pk_max_i16 {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def v_pk_min_i16(self, d, x, y):
        def fn(w):
            p, q = w.rd(x), w.rd(y)
            out = np.zeros(64, dtype=np.uint32)
            for sh in (0, 16):
                pa = ((p >> sh) & 0xffff).astype(np.uint16).view(np.int16)
                qa = ((q >> sh) & 0xffff).astype(np.uint16).view(np.int16)
                out |= np.minimum(pa, qa).view(np.uint16).astype(np.uint32) << sh
            w.wr(d, out)
        self._valu(f"v_pk_min_i16 {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def _pk16(self, name, d, x, y, f, suffix=""):
        def fn(w):
            p, q = w.rd(x), w.rd(y)
            out = np.zeros(64, dtype=np.uint32)
            for sh in (0, 16):
                pa = ((p >> sh) & 0xffff).astype(np.int64)
                qa = ((q >> sh) & 0xffff).astype(np.int64)
                out |= (f(pa, qa) & 0xffff).astype(np.uint32) << sh
            w.wr(d, out)
        self._valu(f"{name} {_txt(d)}, {_txt(x)}, {_txt(y)}{suffix}", fn)

    def v_pk_min_u16(self, d, x, y):
        self._pk16("v_pk_min_u16", d, x, y, np.minimum)

    def v_pk_sub_u16(self, d, x, y, clamp=False):
        """per 16-bit half x - y; clamp: saturate at 0"""
        if clamp:
            self._pk16("v_pk_sub_u16", d, x, y, lambda p, q: np.maximum(p - q, 0), " clamp")
        else:
            self._pk16("v_pk_sub_u16", d, x, y, lambda p, q: p - q)

    def v_bfe_i32(self, d, x, off, width):
        def fn(w):
            p = w.rd(x).astype(np.int64)
            o, wd = w.rd(off).astype(np.int64) & 31, w.rd(width).astype(np.int64) & 31
            val = (p >> o) & ((1 << wd) - 1)
            sign = (val >> (wd - 1)) & 1
            val = np.where(sign == 1, val - (1 << wd), val)
            w.wr(d, (val & 0xffffffff).astype(np.uint32))
        self._valu(f"v_bfe_i32 {_txt(d)}, {_txt(x)}, {_txt(off)}, {_txt(width)}", fn)

    def v_bfi_b32(self, d, m, x, y):
        """(m & x) | (~m & y)"""
        self._v3("v_bfi_b32", d, m, x, y, lambda p, q, r: (p & q) | ((p ^ 0xffffffff) & r))

    def v_permlane32_swap(self, x, y):
        """swap x[lanes 32..63] with y[lanes 0..31] (a VALU write of x or y needs a wait state before it: padded here)"""
        self.s_nop(1)

        def fn(w):
            a_, b_ = w.rd(x), w.rd(y)
            na, nb_ = a_.copy(), b_.copy()
            na[32:] = b_[:32]
            nb_[:32] = a_[32:]
            w.wr(x, na, masked=False)
            w.wr(y, nb_, masked=False)
        self._valu(f"v_permlane32_swap_b32 {_txt(x)}, {_txt(y)}", fn)

    def v_dot2c_f32_bf16(self, d, x, y):
        """d += x.bf16[0] * y.bf16[0] + x.bf16[1] * y.bf16[1] (fp32)"""
        def fn(w):
            w.in_dot = True
            p, q = w.rd(x), w.rd(y)
            acc = w.rd(d).view(np.float32).astype(np.float64)
            w.in_dot = False
            for sh in (0, 16):
                acc = acc + bf16_to_f32(((p >> sh) & 0xffff).astype(np.uint16)).astype(np.float64) * bf16_to_f32(((q >> sh) & 0xffff).astype(np.uint16)).astype(np.float64)
            w.wr(d, acc.astype(np.float32).view(np.uint32))
            w.dotw[d.idx] = w.nstep
        self._valu(f"v_dot2c_f32_bf16 {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def v_cmp_u32(self, op, d, x, y):
        """d: VCC or an SGPR pair (e64)"""
        f = {"ge": np.greater_equal, "lt": np.less, "eq": np.equal, "ne": np.not_equal, "gt": np.greater, "le": np.less_equal}[op]

        def fn(w):
            m = f(w.rd(x), w.rd(y)) & w.lanes()
            val = 0
            for i in np.nonzero(m)[0]:
                val |= 1 << int(i)
            w.swr(d, val & 0xffffffff, 0)
            w.swr(d, val >> 32, 1)
        self._valu(f"v_cmp_{op}_u32 {_txt(d)}, {_txt(x)}, {_txt(y)}", fn)

    def v_cndmask_b32(self, d, x, y, m):
        """d = mask bit ? y : x"""
        def fn(w):
            val = w.srd(m, 0) | (w.srd(m, 1) << 32)
            bits = np.array([(val >> i) & 1 for i in range(64)], dtype=bool)
            w.wr(d, np.where(bits, w.rd(y), w.rd(x)))
        self._valu(f"v_cndmask_b32 {_txt(d)}, {_txt(x)}, {_txt(y)}, {_txt(m)}", fn)

    # ------------------------------------------------------------------ MFMA
    def v_mfma_f32_32x32x16_bf16(self, d, sa, sb, c):
        """D(32x32) = A(32x16) . B(16x32) + C.  lane l: A[l&31][8(l>>5)+e], B[8(l>>5)+e][l&31]; D[r]: col l&31, row (r&3) + 8(r>>2) + 4(l>>5)."""
        assert d.n == 16 and sa.n == 4 and sb.n == 4

        def fn(w):
            def frag(r):
                m = np.zeros((32, 16), dtype=np.float32)
                for q in range(4):
                    u = w.rd(r, q)
                    for half in range(2):
                        val = bf16_to_f32(((u >> (16 * half)) & 0xffff).astype(np.uint16))
                        e = 2 * q + half
                        for h in range(2):
                            m[:, 8 * h + e] = val[32 * h:32 * h + 32]
                return m
            A_ = frag(sa).astype(np.float64)          # [i][k]
            B_ = frag(sb).astype(np.float64)          # [j][k]
            P = A_ @ B_.T                             # [i][j]
            for r in range(16):
                rows = (r & 3) + 8 * (r >> 2) + 4 * (LANES >> 5)
                cols = LANES & 31
                cin = np.zeros(64, dtype=np.float32) if (isinstance(c, int) and c == 0) else w.rd(c, r).view(np.float32)
                with np.errstate(all="ignore"):
                    out = (P[rows, cols] + cin.astype(np.float64)).astype(np.float32)
                w.wr(d, out.view(np.uint32), r, masked=False)
        self._add(f"v_mfma_f32_32x32x16_bf16 {_txt(d)}, {_txt(sa)}, {_txt(sb)}, {_txt(c)}", fn, "mfma")

    # ------------------------------------------------------------------ LDS
    def _ds_addr(self, w, addr, off):
        return w.rd(addr).astype(np.int64) + off

    def ds_read(self, d, addr, off=0):
        nb = 4 * d.n
        name = {4: "ds_read_b32", 8: "ds_read_b64", 16: "ds_read_b128"}[nb]
        assert 0 <= off <= 65535

        def fn(w):
            ad = self._ds_addr(w, addr, off)
            lds = w.emu.lds
            if np.any(ad[w.lanes()] % (8 if nb == 8 else nb if nb < 16 else 16)) or np.any(ad[w.lanes()] + nb > lds.size):
                raise EmuError(f"wave {w.wid}: misaligned / out-of-range LDS read: {name} off {off}")
            data = np.stack([lds[np.clip(ad + b, 0, lds.size - 1)] for b in range(nb)], axis=1)   # [64][nb]
            w.emu.lds_rd_bytes += nb * 64
            w._chk(d, "load-destination write")
            w.mark(d)
            m = w.lanes()

            def land():
                w.unmark(d)
                words = data.reshape(64, d.n, 4).astype(np.uint32)
                for i in range(d.n):
                    val = words[:, i, 0] | (words[:, i, 1] << 8) | (words[:, i, 2] << 16) | (words[:, i, 3] << 24)
                    arr = w.V if d.kind == "v" else w.A
                    arr[d.idx + i][m] = val[m]
            w.lgq.append(land)
            if len(w.lgq) > 15:
                raise EmuError(f"wave {w.wid}: more than 15 LGKM operations outstanding (4-bit counter)")
        self._add(f"{name} {_txt(d)}, {_txt(addr)}" + (f" offset:{off}" if off else ""), fn, "lds")

    def ds_read_b64_tr_b16(self, d, addr, off=0):
        """gfx950 transposed LDS read: within each group of 16 lanes, lane p addresses 4 contiguous bf16 = row p >> 2, column quad p & 3 of a
        [4][16] block; lane i receives column i of the block (rows 0..3) as 4 bf16 = 2 dwords."""
        assert d.n == 2 and 0 <= off <= 65535

        def fn(w):
            ad = self._ds_addr(w, addr, off)
            lds = w.emu.lds
            if np.any(ad % 8) or np.any(ad + 8 > lds.size):
                raise EmuError(f"wave {w.wid}: misaligned / out-of-range transposed LDS read")
            raw = np.stack([lds[ad + b] for b in range(8)], axis=1).astype(np.uint32)       # [64][8]
            h = (raw[:, 0::2] | (raw[:, 1::2] << 8))                                         # [64][4] bf16 of the addressed quad
            out = np.zeros((64, 4), dtype=np.uint32)
            for l in range(64):
                g, i = l & ~15, l & 15
                for r in range(4):
                    out[l, r] = h[g + 4 * r + (i >> 2), i & 3]
            w.emu.lds_rd_bytes += 8 * 64
            w._chk(d, "load-destination write")
            w.mark(d)

            def land():
                w.unmark(d)
                arr = w.V if d.kind == "v" else w.A
                arr[d.idx] = out[:, 0] | (out[:, 1] << 16)
                arr[d.idx + 1] = out[:, 2] | (out[:, 3] << 16)
            w.lgq.append(land)
            if len(w.lgq) > 15:
                raise EmuError(f"wave {w.wid}: more than 15 LGKM operations outstanding (4-bit counter)")
        self._add(f"ds_read_b64_tr_b16 {_txt(d)}, {_txt(addr)}" + (f" offset:{off}" if off else ""), fn, "lds")

    def ds_write(self, addr, src, off=0):
        nb = 4 * src.n
        name = {4: "ds_write_b32", 8: "ds_write_b64", 16: "ds_write_b128"}[nb]
        assert 0 <= off <= 65535

        def fn(w):
            ad = self._ds_addr(w, addr, off)
            lds = w.emu.lds
            m = w.lanes()
            if np.any(ad[m] % min(nb, 16 if nb == 16 else nb)) or np.any(ad[m] + nb > lds.size):
                raise EmuError(f"wave {w.wid}: misaligned / out-of-range LDS write")
            for i in range(src.n):
                val = w.rd(src, i)
                for b in range(4):
                    lds[(ad + 4 * i + b)[m]] = ((val >> (8 * b)) & 0xff).astype(np.uint8)[m]
            w.lgq.append(lambda: None)
            if len(w.lgq) > 15:
                raise EmuError(f"wave {w.wid}: more than 15 LGKM operations outstanding (4-bit counter)")
        self._add(f"{name} {_txt(addr)}, {_txt(src)}" + (f" offset:{off}" if off else ""), fn, "lds")

    # ------------------------------------------------------------------ buffer (MUBUF) instructions, raw descriptors (stride 0), offen
    @staticmethod
    def _buf_addr(w, vaddr, srd, soff, off):
        base = w.srd(srd, 0) | ((w.srd(srd, 1) & 0xffff) << 32)
        nrec = w.srd(srd, 2)
        voff = w.rd(vaddr).astype(np.int64) if vaddr is not None else np.zeros(64, dtype=np.int64)
        so = w.srd(soff) if isinstance(soff, Reg) else soff
        return base + so + voff + off, voff + off, nrec

    def buffer_load(self, d, vaddr, srd, soff, off=0, nt=False):
        nd = d.n
        name = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 4: "buffer_load_dwordx4"}[nd]
        assert 0 <= off <= 4095

        def fn(w):
            ad, rng, nrec = self._buf_addr(w, vaddr, srd, soff, off)
            m = w.lanes()
            vals = np.zeros((64, nd), dtype=np.uint32)
            for l in np.nonzero(m)[0]:
                if rng[l] + 4 * nd > nrec:
                    continue        # out of range: returns 0
                vals[l] = w.emu.mem_read(int(ad[l]), 4 * nd).view(np.uint32)
            w._chk(d, "load-destination write")
            w.mark(d)
            w.emu.vmem_ld_bytes += 4 * nd * int(m.sum())

            def land():
                w.unmark(d)
                arr = w.V if d.kind == "v" else w.A
                for i in range(nd):
                    arr[d.idx + i][m] = vals[m, i]
            w.vmq.append(land)
            w.retire(w.vmq, 63)          # 6-bit counter: the hardware stalls the issue until an older operation has returned
        self._add(f"{name} {_txt(d)}, {_txt(vaddr)}, {_txt(srd)}, {_txt(soff)} offen" + (f" offset:{off}" if off else ""), fn, "vmem")

    def buffer_load_lds_x4(self, vaddr, srd, soff):
        """buffer_load_dwordx4 ... lds: lane i's 16 bytes land at LDS[M0 + 16 i] (lane-linear), when this wave retires the operation."""
        def fn(w):
            ad, rng, nrec = self._buf_addr(w, vaddr, srd, soff, 0)
            m = w.lanes()
            dst0 = w.m0
            if dst0 % 16 or dst0 + 1024 > w.emu.lds.size:
                raise EmuError(f"LDS-DMA destination {dst0}")
            data = np.zeros((64, 16), dtype=np.uint8)
            for l in np.nonzero(m)[0]:
                if rng[l] + 16 <= nrec:
                    data[l] = w.emu.mem_read(int(ad[l]), 16)
            # poison now (the bytes are undefined until the DMA lands): bf16 NaNs
            lds = w.emu.lds
            pois = np.tile(np.array([BF16_NAN & 0xff, BF16_NAN >> 8], dtype=np.uint8), 8)
            for l in np.nonzero(m)[0]:
                lds[dst0 + 16 * l: dst0 + 16 * l + 16] = pois
            w.emu.vmem_ld_bytes += 16 * int(m.sum())

            def land():
                for l in np.nonzero(m)[0]:
                    lds[dst0 + 16 * l: dst0 + 16 * l + 16] = data[l]
            w.vmq.append(land)
            w.retire(w.vmq, 63)          # 6-bit counter: the hardware stalls the issue until an older operation has returned
        self._add(f"buffer_load_dwordx4 {_txt(vaddr)}, {_txt(srd)}, {_txt(soff)} offen lds", fn, "vmem")

    def global_load_lds_x4(self, voff, sbase, mods=""):
        """global_load_lds_dwordx4 voff, s[base:base+1]: lane i fetches 16 bytes at sbase + voff[i] and they land at LDS[M0 + 16 i] when this
        wave retires the operation (the form the HIP kernels of csrc/gemm.hip use through __builtin_amdgcn_global_load_lds)."""
        def fn(w):
            base = w.srd(sbase, 0) | (w.srd(sbase, 1) << 32)
            ad = base + w.rd(voff).astype(np.int64)
            m = w.lanes()
            dst0 = w.m0
            if dst0 % 16 or dst0 + 1024 > w.emu.lds.size:
                raise EmuError(f"LDS-DMA destination {dst0}")
            data = np.zeros((64, 16), dtype=np.uint8)
            for l in np.nonzero(m)[0]:
                data[l] = w.emu.mem_read(int(ad[l]), 16)
            lds = w.emu.lds
            pois = np.tile(np.array([BF16_NAN & 0xff, BF16_NAN >> 8], dtype=np.uint8), 8)
            for l in np.nonzero(m)[0]:
                lds[dst0 + 16 * l: dst0 + 16 * l + 16] = pois
            w.emu.vmem_ld_bytes += 16 * int(m.sum())

            def land():
                for l in np.nonzero(m)[0]:
                    lds[dst0 + 16 * l: dst0 + 16 * l + 16] = data[l]
            w.vmq.append(land)
            w.retire(w.vmq, 63)
        self._add(f"global_load_lds_dwordx4 {_txt(voff)}, {_txt(sbase)}" + (f" {mods}" if mods else ""), fn, "vmem")

    def buffer_atomic_add_f32(self, src, vaddr, srd, soff, off=0):
        """no-return fp32 atomic add of src (one dword per lane)"""
        assert src.n == 1 and 0 <= off <= 4095

        def fn(w):
            ad, rng, nrec = self._buf_addr(w, vaddr, srd, soff, off)
            m = w.lanes()
            vals = w.rd(src).view(np.float32).copy()

            def land():
                for l in np.nonzero(m)[0]:
                    if rng[l] + 4 <= nrec:
                        _atomic_add_f32(w.emu, int(ad[l]), vals[l])
            w.vmq.append(land)
            w.retire(w.vmq, 63)
        self._add(f"buffer_atomic_add_f32 {_txt(src)}, {_txt(vaddr)}, {_txt(srd)}, {_txt(soff)} offen" + (f" offset:{off}" if off else ""), fn, "vmem")

    def buffer_store(self, src, vaddr, srd, soff, off=0, nt=False):
        nd = src.n
        name = {1: "buffer_store_dword", 2: "buffer_store_dwordx2", 4: "buffer_store_dwordx4"}[nd]
        assert 0 <= off <= 4095

        def fn(w):
            ad, rng, nrec = self._buf_addr(w, vaddr, srd, soff, off)
            m = w.lanes()
            vals = np.stack([w.rd(src, i) for i in range(nd)], axis=1)
            w.emu.vmem_st_bytes += 4 * nd * int(m.sum())

            def land():
                for l in np.nonzero(m)[0]:
                    if rng[l] + 4 * nd > nrec:
                        continue
                    w.emu.mem_write(int(ad[l]), vals[l].view(np.uint8))
            w.vmq.append(land)
            w.retire(w.vmq, 63)          # 6-bit counter: the hardware stalls the issue until an older operation has returned
        self._add(f"{name} {_txt(src)}, {_txt(vaddr)}, {_txt(srd)}, {_txt(soff)} offen" + (f" offset:{off}" if off else "") + (" nt" if nt else ""), fn, "vmem")


def _atomic_add_f32(emu, addr, val):
    cur = emu.mem_read(addr, 4).view(np.float32)[0]
    emu.mem_write(addr, np.array([np.float32(cur) + np.float32(val)], dtype=np.float32).view(np.uint8))


class Emu:
    """Runs one workgroup (nwaves waves) of a Prog.  Memory = named numpy byte buffers at fake device addresses."""

    def __init__(self, prog, lds_bytes=163840, nwaves=4):
        self.prog = prog
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        rs = np.random.RandomState(7)
        self.lds[:] = rs.randint(0, 256, size=lds_bytes)
        self.bufs = []       # (base, ndarray uint8, writable)
        self.next_base = 0x7f0000000000
        self.nwaves = nwaves
        self.lds_rd_bytes = 0
        self.vmem_ld_bytes = 0
        self.vmem_st_bytes = 0

    def alloc(self, arr, writable=False):
        """register a numpy array (viewed as bytes, shares memory) and return its device address"""
        b = arr.view(np.uint8).reshape(-1)
        base = self.next_base
        self.bufs.append((base, b, writable))
        self.next_base += (b.size + 0xffff) & ~0xfff
        return base

    def _find(self, addr, n):
        for base, b, wr in self.bufs:
            if base <= addr and addr + n <= base + b.size:
                return b, addr - base, wr
        raise EmuError(f"memory access fault: address 0x{addr:x} (+{n}) is outside every buffer")

    def mem_read(self, addr, n):
        b, o, _ = self._find(addr, n)
        return b[o:o + n].copy()

    def mem_write(self, addr, data):
        b, o, wr = self._find(addr, data.size)
        if not wr:
            raise EmuError(f"write to read-only buffer at 0x{addr:x}")
        b[o:o + data.size] = data

    def run(self, kernarg, wg_id, order=None, max_steps=50_000_000, wg_id_y=0):
        """kernarg: bytes.  s[0:1] = kernarg address, s2 = workgroup id x, s3 = workgroup id y (kernels that enable it), v0 = thread id in the workgroup."""
        ka = np.frombuffer(bytes(kernarg), dtype=np.uint8).copy()
        kaddr = self.alloc(ka)
        waves = [Wave(i, self) for i in range(self.nwaves)]
        for w in waves:
            w.S[0] = kaddr & 0xffffffff
            w.S[1] = kaddr >> 32
            w.S[2] = wg_id
            w.S[3] = wg_id_y
            w.V[0] = (np.arange(64) + 64 * w.wid).astype(np.uint32)
        order = list(order) if order is not None else list(range(self.nwaves))
        ops = self.prog.ops
        steps = 0
        while not all(w.done for w in waves):
            progressed = False
            for wi in order:
                w = waves[wi]
                while not w.done and not w.at_barrier:
                    op = ops[w.pc]
                    if op.fn is not None:
                        w.nstep += 1
                        op.fn(w)
                        w.counts[op.kind] = w.counts.get(op.kind, 0) + 1
                    w.pc += 1
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise EmuError("step limit")
            live = [w for w in waves if not w.done]
            if live and all(w.at_barrier for w in live):
                if len(live) != len(waves):
                    raise EmuError("barrier reached by some waves after others ended")
                for w in live:
                    w.at_barrier = False
                progressed = True
            if not progressed:
                raise EmuError("deadlock")
        return waves
