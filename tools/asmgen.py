"""Instruction-stream bookkeeping shared by the kernel generators (tools/gen_ffn_*.py; tools/gen_attn_fwd4.py carries its own older copy).

A generated kernel body is ONE inline-asm block with fixed registers.  What the hardware does not interlock (or what costs a full drain when
left to `s_waitcnt lgkmcnt(0)`) is tracked here, per wave, in program order:
  * the LDS queue: reads AND writes count in lgkmcnt and retire in order; before an instruction touches the destination of a pending
    read, `s_waitcnt lgkmcnt(N)` with N = the operations issued after that read;
  * the vector-memory queue (loads, stores, LDS-DMA: one counter, in order): the caller names what has to be done (`vm_wait_for(tag)`);
  * MFMA results: a VALU / memory instruction that reads or writes a register an 8-pass MFMA wrote needs 12 wait states in between
    (hipcc emits `s_nop 11` for the same case); independent instructions in between count, the rest is padded with s_nop.
"""


def v(r, n=1):
    return "v%d" % r if n == 1 else "v[%d:%d]" % (r, r + n - 1)


def a(r, n=1):
    return "a%d" % r if n == 1 else "a[%d:%d]" % (r, r + n - 1)


def s(r, n=1):
    return "s%d" % r if n == 1 else "s[%d:%d]" % (r, r + n - 1)


def rng(base, n):
    return list(range(base, base + n))


class Stream:
    MFMA_WAIT = 13      # instructions between an 8-pass MFMA and a VALU / memory instruction that touches its result (12 states + 1)

    def __init__(self):
        self.out = []
        self.n = 0                  # wait states issued so far
        self.mfma_w = {}            # register name ("v12" / "a3") -> wait-state index of the MFMA that last wrote it
        self.lds = []               # pending LDS operations, oldest first: sets of destination registers (empty set: a write)
        self.vm = []                # pending vector-memory operations, oldest first: tags
        self.nops = 0
        self.counts = {}

    # ---- raw text ------------------------------------------------------------------------------------------------------------
    def raw(self, txt, states=1, kind="misc"):
        self.out.append(txt)
        self.n += states
        self.counts[kind] = self.counts.get(kind, 0) + 1

    def comment(self, txt):
        self.out.append("; " + txt)

    def label(self, name):
        self.out.append(name + ":")

    # ---- hazards -------------------------------------------------------------------------------------------------------------
    def _need(self, regs):
        last = -1
        for i, dst in enumerate(self.lds):
            if dst & regs:
                last = i
        if last >= 0:
            left = len(self.lds) - 1 - last
            assert left <= 15, "lgkmcnt field is 4 bits"
            self.raw("s_waitcnt lgkmcnt(%d)" % left, kind="wait")
            self.lds = self.lds[last + 1:]

    def _mfma_pad(self, regs):
        need = 0
        for r in regs:
            if r in self.mfma_w:
                need = max(need, self.mfma_w[r] + self.MFMA_WAIT - self.n)
        while need > 0:
            k = min(need, 8)
            self.raw("s_nop %d" % (k - 1), states=k, kind="nop")
            self.nops += k
            need -= k

    def valu(self, txt, reads=(), writes=(), kind="valu"):
        regs = set(reads) | set(writes)
        self._need(regs)
        self._mfma_pad(regs)
        self.raw(txt, kind=kind)
        for r in writes:
            self.mfma_w.pop(r, None)

    def salu(self, txt):
        self.raw(txt, kind="salu")

    def mfma(self, txt, ab_reads, c_reads, writes):
        """ab_reads: the A / B operand registers; c_reads: SrcC when it is not the destination itself (a chain on one accumulator
        issues back to back); writes: the destination."""
        self._need(set(ab_reads) | set(c_reads))
        self._mfma_pad(set(ab_reads) | (set(c_reads) - set(writes)))
        self.raw(txt, kind="mfma")
        for r in writes:
            self.mfma_w[r] = self.n - 1

    def lds_read(self, txt, addr, writes):
        self._need(set(addr) | set(writes))
        self._mfma_pad(set(addr) | set(writes))
        self.raw(txt, kind="lds")
        self.lds.append(set(writes))

    def lds_write(self, txt, reads):
        self._need(set(reads))
        self._mfma_pad(set(reads))
        self.raw(txt, kind="ldsw")
        self.lds.append(set())

    def drain_lds(self, vm=None):
        self.raw("s_waitcnt lgkmcnt(0)" if vm is None else "s_waitcnt vmcnt(%d) lgkmcnt(0)" % vm, kind="wait")
        self.lds = []

    def vmem(self, txt, reads=(), writes=(), tag="vm", kind="vmem"):
        """a load (writes = its destination registers), a store (reads = address + data) or an LDS-DMA request"""
        regs = set(reads) | set(writes)
        self._need(regs)
        self._mfma_pad(regs)
        self.raw(txt, kind=kind)
        self.vm.append(tag)

    def vm_left_after(self, tags):
        """vmcnt value that leaves in flight only what was issued after the last operation carrying one of `tags`"""
        last = -1
        for i, t in enumerate(self.vm):
            if t in tags:
                last = i
        return len(self.vm) - 1 - last

    def vm_wait(self, tags, with_lds=False):
        """wait until every pending vector-memory operation up to the last one tagged in `tags` is done"""
        left = self.vm_left_after(tags)
        if left == len(self.vm) and not with_lds:
            return
        assert left <= 63
        if with_lds:
            self.raw("s_waitcnt vmcnt(%d) lgkmcnt(0)" % left, kind="wait")
            self.lds = []
        else:
            self.raw("s_waitcnt vmcnt(%d)" % left, kind="wait")
        self.vm = self.vm[len(self.vm) - left:] if left else []

    # ---- loops: the state at a back edge has to equal the state at the loop's entry ------------------------------------------------
    def snapshot(self):
        return dict(pos=len(self.out), n=self.n, mfma=dict(self.mfma_w), lds=[set(x) for x in self.lds], vm=list(self.vm))

    def rewind(self, snap, seed_from_now=True):
        """back to `snap`, keeping (shifted) the MFMA write times the body left behind: the second pass of a loop body then pads for
        results of the previous trip"""
        shift = self.n - snap["n"]
        seeded = {r: w - shift for r, w in self.mfma_w.items()} if seed_from_now else {}
        del self.out[snap["pos"]:]
        self.n = snap["n"]
        self.mfma_w = dict(snap["mfma"])
        for r, w in seeded.items():
            self.mfma_w[r] = max(w, self.mfma_w.get(r, -10 ** 9))
        self.lds = [set(x) for x in snap["lds"]]
        self.vm = list(snap["vm"])
        self.nops = 0
        self.counts = {}


def write_inc(path, header, macros, clobber_name, clobbers):
    with open(path, "w") as f:
        f.write("// %s - do not edit (edit the generator and run it again)\n" % header)
        for name, lines in macros:
            f.write("#define %s \\\n" % name)
            for line in lines:
                f.write('    "%s\\n" \\\n' % line.replace("\\", "\\\\").replace('"', '\\"'))
            f.write('    ""\n')
        f.write("#define %s %s\n" % (clobber_name, ", ".join('"%s"' % r for r in clobbers)))
