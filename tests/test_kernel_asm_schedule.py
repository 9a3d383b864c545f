"""The hand-scheduled k-steps of k_vara_i8p / k_syrk_f4p (csrc/eagle_i8mfma.hip) as text: an in-order model of the LDS return
queue run over the instruction sequence of every k-step macro.  LDS data returns in issue order and the kernels wait with
COUNTED `s_waitcnt lgkmcnt(N)`; a wrong N, an MFMA placed before its fragment's wait, or a re-load issued before the last MFMA
that reads the register would give silently wrong sums only when the timing is unlucky.  Checked here without a GPU:
  * every MFMA operand has returned (worst case: nothing returns before a wait forces it);
  * a fragment register is never re-loaded while a later MFMA of the same k-step still reads it;
  * the queue a k-step leaves behind is, name for name, the queue the next k-step's waits assume."""
import os
import re

from conftest import ROOT

# (round 4: the 384 x 256 engine's macros moved to csrc/eagle_t8.h, shared with the int8 W products of eagle_w8.hip)
SRC = open(os.path.join(ROOT, "eagleeverything_amd", "csrc", "eagle_t8.h")).read() + open(os.path.join(ROOT, "eagleeverything_amd", "csrc", "eagle_i8mfma.hip")).read()


def macro_body(name):
    m = re.search(r"#define %s(\([^)]*\))?\s*\\?\n((?:.*\\\n)*.*\n)" % re.escape(name), SRC)
    assert m, name
    return m.group(2)


TOKEN = re.compile(r'(X_WT|X_LD|X_DM|X_MF|X_MZ|S_MF|S_MZ|\bM)\(([^)]*)\)|(s_waitcnt vmcnt\(0\) lgkmcnt\(0\))|(s_barrier)|\b(DA[123])\b')


def run(body, entry, loads_expected):
    """entry: outstanding loads on entry, oldest first.  Returns (exit queue, list of (kind, args))."""
    queue = list(entry)
    reloaded = set()
    seq = []
    for t in TOKEN.finditer(body):
        if t.group(3):
            queue.clear()
            seq.append(("wait_all",))
            continue
        if t.group(4):
            assert not queue, "barrier with LDS reads of this buffer outstanding"
            seq.append(("barrier",))
            continue
        if t.group(5):   # a DMA slot of a generic macro (its operand set is the instantiation's business)
            seq.append(("dma", t.group(5)))
            continue
        kind, args = t.group(1), [a.strip() for a in t.group(2).split(",")]
        if kind == "X_WT":
            n = int(args[0])
            while len(queue) > n:
                queue.pop(0)
        elif kind == "X_LD":
            dst = args[0]
            assert dst not in queue, "%s loaded twice" % dst
            queue.append(dst)
            reloaded.add(dst)
        elif kind in ("M", "X_MF", "X_MZ", "S_MF", "S_MZ"):
            c, a, b = args
            for r in (a, b):
                assert r not in queue, "MFMA %s reads %s before its load is known to have returned (queue %s)" % (c, r, queue)
                assert r not in reloaded, "MFMA %s reads %s after its re-load for the next k-step was issued" % (c, r)
            seq.append(("mfma", c, a, b))
        elif kind == "X_DM":
            seq.append(("dma",) + tuple(args))
    assert reloaded == set(loads_expected), (sorted(reloaded), sorted(loads_expected))
    return queue, seq


def rename(queue, mapping):
    return [mapping.get(r, r) for r in queue]


def test_vara_kstep_queue_discipline():
    entry = ["a0", "a1", "b0", "a2", "b1", "b2", "b3"]          # issue order of the previous k-step's loads, as this k-step names them
    x_to_a = {"x0": "a0", "x1": "a1", "x2": "a2"}                # the double-buffered genotype fragments swap roles every k-step
    body = macro_body("X_KSTEP")
    leaves, seq = run(body, entry, ["x0", "x1", "x2", "b0", "b1", "b2", "b3"])
    assert rename(leaves, x_to_a) == entry
    mf = [s for s in seq if s[0] == "mfma"]
    assert len(mf) == 12 and len({s[1] for s in mf}) == 12       # 3 x 4 tiles, each once
    assert {(s[2], s[3]) for s in mf} == {("a%d" % m, "b%d" % n) for m in range(3) for n in range(4)}
    assert all(s[1] == "c%s%s" % (s[2][1], s[3][1]) for s in mf)  # accumulator c<m><n> gets a<m> x b<n>
    last, seq = run(macro_body("X_KLAST_G"), entry, ["x0", "x1", "x2", "b0", "b1", "b2", "b3"])
    assert rename(last, x_to_a) == entry
    kinds = [s[0] for s in seq]
    assert kinds.index("wait_all") < kinds.index("barrier") < kinds.index("dma")   # DMA into the freed buffer only behind the barrier
    assert sum(k == "mfma" for k in kinds) == 12 and sum(k == "dma" for k in kinds) == 3
    # the DMA slots of a stage: 3 genotype loads behind the barrier + (3 genotype, 3 digit) + 1 digit = 6 + 4 row groups per wave
    assert [s[1] for s in seq if s[0] == "dma"] == ["DA1", "DA2", "DA3"]
    inst = re.search(r"#define X_KLAST X_KLAST_G\((.*)\)\n", SRC).group(1)
    assert [a.strip() for a in re.findall(r"X_DM\(([^)]*)\)", inst)] == ["vE, a", "vO, a", "vE, a"] and inst.startswith("X_MF")
    # and the 384 x 256 MM^T kernel instantiates the same macro with the fp4 instruction and the same DMA slots
    inst = re.search(r"asm volatile\(X_KLAST_G\((.*?)\)\n", SRC).group(1)
    assert [a.strip() for a in re.findall(r"X_DM\(([^)]*)\)", inst)] == ["vE, a", "vO, a", "vE, a"] and inst.startswith("S_MF")


def test_syrk_kstep_queue_discipline():
    entry = ["a0", "b0", "a1", "b1", "a2", "a3"]
    nxt = {"x0": "a0", "y0": "b0", "x1": "a1", "y1": "b1", "x2": "a2", "x3": "a3"}
    leaves, seq = run(macro_body("S_KSTEP"), entry, list(nxt))
    assert rename(leaves, nxt) == entry
    mf = [s for s in seq if s[0] == "mfma"]
    assert len(mf) == 8 and {(s[2], s[3]) for s in mf} == {("a%d" % m, "b%d" % n) for m in range(4) for n in range(2)}
    assert all(s[1] == "c%s%s" % (s[2][1], s[3][1]) for s in mf)
    last, seq = run(macro_body("S_KLAST"), entry, list(nxt))
    assert rename(last, nxt) == entry
    kinds = [s[0] for s in seq]
    assert kinds.index("wait_all") < kinds.index("barrier") < kinds.index("dma")
    assert sum(k == "mfma" for k in kinds) == 8
    assert [s[1:] for s in seq if s[0] == "dma"] == [("vE", "a"), ("vO", "a"), ("vE", "a"), ("vO", "a"), ("vE", "b"), ("vO", "b")]


def test_prologues_issue_in_queue_order():
    """The pipeline-fill loads must enter the queue in the order the first k-step's waits assume (they end in a full wait, but the
    order is part of the contract the k-step macros are written against)."""
    def order(fn):
        m = re.search(r"void %s\(.*?asm volatile\((.*?):" % fn, SRC, re.S)
        assert m, fn
        return [t.group(2).split(",")[0].strip() for t in TOKEN.finditer(m.group(1)) if t.group(1) == "X_LD"]
    assert order("tx_prologue") == ["x0", "x1", "b0", "x2", "b1", "b2", "b3"]
    assert order("sx_prologue") == ["x0", "y0", "x1", "y1", "x2", "x3"]
