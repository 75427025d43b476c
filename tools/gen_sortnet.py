#!/usr/bin/env python3
"""Generate astroburst_amd/csrc/sortnet_gen.hpp: fully unrolled sorting networks for 2..256 inputs whose every register index is a
compile-time constant (a runtime-indexed sample vector would be demoted to scratch memory on gfx950).

The networks are Batcher's odd-even merge sort (543 compare-exchanges for 64 inputs; bitonic: 672), halves first, REWRITTEN over the
instructions gfx950 has: v_min3_f32 / v_med3_f32 / v_max3_f32 issue at the same half rate as a two-input v_min_f32 / v_max_f32
(tools/valu_rate.hip: 4.3 cycles per wave64 instruction, all five), so every value that can be produced by ONE three-input operation
instead of two chained two-input ones is an instruction saved:

  * base case, four unsorted wires: sort three (min3 / med3 / max3), then place the fourth (min, med3, med3, max) -- 7 instructions
    where five compare-exchanges take 10;
  * FUSED compare-exchanges.  Let w = min(X, Y) be the low output of one exchange and let the exchange that consumes it compare it
    with u.  Its outputs are  min(u, w) = min3(u, X, Y)  and  max(u, w) = max(u, min(X, Y)) = med3(u, X, Y)  -- the second identity
    holds iff u <= max(X, Y), which in a merge network is a property of the NETWORK (u and max(X, Y) are monotone functions of the
    inputs), so the 0-1 principle decides it: check u <= max(X, Y) on every 0-1 input that can reach that point.  Then w itself is
    never materialised: two exchanges cost 3 instructions instead of 4, and an exchange BOTH of whose outputs are absorbed this way
    costs nothing (three exchanges in 4 instructions).  Mirror image for the high output: max3(u, X, Y) and med3(u, X, Y) iff
    u >= min(X, Y).  The minimum / maximum of the base case (a plain two-input min / max) can be absorbed in the same way.
    An exchange whose outputs are three-input operations cannot be absorbed in turn, so which exchanges absorb which is a
    combinatorial choice: a 0/1 programme (scipy HiGHS) maximises the number of absorbed values.  64 inputs: 463 exchanges + 16 base
    cases = 1038 instructions as written by Batcher, 789 after the rewrite.

Every rewrite is value preserving on its own, and is CHECKED that way: the generator runs the plain network and the rewritten
programme side by side on every 0-1 input of every merge level (all 4-run states of a block: (s/4 + 1)^4 for block size s, every
0-1 input for s = 8; the validity of a rewrite depends on nothing outside its block) plus random 0-1 vectors, and asserts that every
value the programme materialises equals the wire value of the plain network.

Run:  python tools/gen_sortnet.py            (writes the header; ~1 min, most of it the 128 / 256 networks)
      python tools/gen_sortnet.py --check    (the header on disk is what this generator writes)
"""
import itertools
import os
import sys

import numpy as np

SIZES = (2, 4, 8, 16, 32, 64, 128, 256)
FUSED_SIZES = (8, 16, 32, 64, 128, 256)
PADDED_SIZES = (128, 256)   # sizes that also get sort_fused_n<NREAL>: the same programme with the wires >= NREAL known to be +inf pads


# ---------------------------------------------------------------------------------------------------------------------------
# Batcher's odd-even merge sort
# ---------------------------------------------------------------------------------------------------------------------------
def merge_ces(lo, cnt):
    """compare-exchanges of the odd-even merge of the sorted runs [lo, lo + cnt/2) and [lo + cnt/2, lo + cnt)"""
    ces = []

    def merge(lo, cnt, r):
        step = r * 2
        if step < cnt:
            merge(lo, cnt, step)
            merge(lo + r, cnt, step)
            for i in range(lo + r, lo + cnt - r, step):
                ces.append((i, i + r))
        else:
            ces.append((lo, lo + r))

    merge(lo, cnt, 1)
    return ces


def batcher(n):
    """Batcher's odd-even merge sort, recursive form, n a power of two.

    Same size as Knuth's merge exchange (543 compare-exchanges for n = 64), but the ORDER matters
    here: the two halves are sorted before anything crosses the middle, so the network can start
    on frames 0..15 while the loads of frames 16..63 are still in flight (the loads retire in issue
    order: `s_waitcnt vmcnt(k)`)."""
    ces = []

    def sort(lo, cnt):
        if cnt > 1:
            m = cnt // 2
            sort(lo, m)
            sort(lo + m, m)
            ces.extend(merge_ces(lo, cnt))

    sort(0, n)
    return ces


def check(n, ces):
    # 0-1 principle on a sample + exhaustive for small n
    import random

    def run(v):
        v = list(v)
        for a, b in ces:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
        return v
    if n <= 16:
        for bits in itertools.product((0, 1), repeat=n):
            assert run(bits) == sorted(bits)
    else:
        rnd = random.Random(1)
        for _ in range(2000):
            bits = [rnd.randint(0, 1) for _ in range(n)]
            assert run(bits) == sorted(bits)
        for _ in range(500):
            k = rnd.randint(0, n)
            bits = [1] * k + [0] * (n - k)
            rnd.shuffle(bits)
            assert run(bits) == sorted(bits)


# ---------------------------------------------------------------------------------------------------------------------------
# the network as a list of items in execution order:  ('S4', lo)  |  ('CE', a, b, size, lo) (size / lo: the merge it belongs to)
# ---------------------------------------------------------------------------------------------------------------------------
def items_of(n):
    items = []

    def sort(lo, cnt):
        if cnt == 4:
            items.append(("S4", lo))
            return
        sort(lo, cnt // 2)
        sort(lo + cnt // 2, cnt // 2)
        for (a, b) in merge_ces(lo, cnt):
            items.append(("CE", a, b, cnt, lo))

    sort(0, n)
    return items


# ---------------------------------------------------------------------------------------------------------------------------
# 0-1 state sets, bit-packed (one bit per state): min = AND, max = OR, med3 = majority  (0 = small: a sorted run is 0 .. 0 1 .. 1)
# ---------------------------------------------------------------------------------------------------------------------------
def level_states(n, size, rng):
    """inputs of the whole n-wire network in which EVERY block of `size` wires runs through all states of its level context:
    size == 8: all 256 0-1 inputs of the block; otherwise all (size/4 + 1)^4 states made of four sorted runs (sorted runs pass
    the lower levels unchanged, so they arrive at the block's two child merges as they are).  Block 0 takes the states in
    order; the other blocks take the same set in a random order each.  Returns uint8 [n, S/8] (bit-packed along the states)."""
    chunks = list(level_state_chunks(n, size, rng))
    assert len(chunks) == 1
    return chunks[0]


CHUNK = 1 << 20   # states per batch of the top level of the 256-wire network (65^4 = 17.9 M states: 18 batches of 32 MB)


QUICK = False   # the test suite's setting: of the 256-wire network's top level, two batches (2 M of 17.9 M states) instead of all


def level_state_chunks(n, size, rng):
    """level_states in batches: one batch for every level but the 256-wire one, whose (64 + 1)^4 states come CHUNK at a time (a
    single block spans the network there, so no permutation of the set is needed: consecutive ranges of the state index)."""
    if size == 8:
        st = np.array(list(itertools.product((0, 1), repeat=8)), dtype=np.uint8)  # [256, 8]
        sets = [st]
    else:
        q = size // 4
        total = (q + 1) ** 4
        pos = np.arange(q, dtype=np.int16)

        def states(first, last):
            idx = np.arange(first, last, dtype=np.int64)
            z = np.stack([(idx // (q + 1) ** (3 - r)) % (q + 1) for r in range(4)], axis=1).astype(np.int16)  # [S, 4], as itertools.product orders them
            return (pos[None, None, :] >= z[:, :, None]).reshape(len(z), size).astype(np.uint8)

        if total > CHUNK and n == size:
            firsts = list(range(0, total, CHUNK))
            if QUICK:
                firsts = [firsts[0], firsts[len(firsts) // 2]]
            sets = (states(f, min(f + CHUNK, total)) for f in firsts)
        else:
            sets = [states(0, total)]
    for st in sets:
        S = len(st)
        cols = []
        for b in range(n // size):
            perm = np.arange(S) if b == 0 else rng.permutation(S)
            cols.append(st[perm])
        full = np.concatenate(cols, axis=1)  # [S, n]
        yield np.packbits(full.T, axis=1)    # [n, ceil(S/8)]


def random_states(n, count, rng):
    full = (rng.random((count, n)) < rng.random((count, 1))).astype(np.uint8)
    return np.packbits(full.T, axis=1)


def maj(a, b, c):
    return (a & b) | (b & c) | (a & c)


# ---------------------------------------------------------------------------------------------------------------------------
# candidate rewrites and their validity
# ---------------------------------------------------------------------------------------------------------------------------
class Net:
    """the plain network of n wires as a DAG of two-output nodes.  Node kinds:
         'CE'  (a, b): X = wire a, Y = wire b -> lo on a, hi on b
         'S4'  lo: the base case; its outer outputs are the virtual exchanges ('S4lo': lo output only = min(s0, x3),
               'S4hi': hi output only = max(s2, x3)) -- modelled as producer nodes with one output each."""

    def __init__(self, n):
        self.n = n
        self.items = items_of(n)
        # producers: for every CE input, who produced it: (node index, 'lo' | 'hi') or None (a three-input output of a base case / input)
        self.nodes = []   # dicts
        prod = {}
        for it in self.items:
            if it[0] == "S4":
                lo = it[1]
                k = len(self.nodes)
                self.nodes.append(dict(kind="S4", lo=lo))
                prod[lo] = (k, "lo")
                prod[lo + 1] = None
                prod[lo + 2] = None
                prod[lo + 3] = (k, "hi")
            else:
                _, a, b, size, lo = it
                k = len(self.nodes)
                self.nodes.append(dict(kind="CE", a=a, b=b, size=size, lo=lo, pa=prod.get(a), pb=prod.get(b)))
                prod[a] = (k, "lo")
                prod[b] = (k, "hi")

    def simulate(self, st, visit):
        """run the plain network on bit-packed states st [n, W]; visit(k, node, X, Y, lo, hi) for every node in order (for a base
        case X / Y are the pair (s0, x3) for its lo output and visit is called twice: once per virtual exchange)."""
        val = [st[w] for w in range(self.n)]
        for k, nd in enumerate(self.nodes):
            if nd["kind"] == "S4":
                lo = nd["lo"]
                x0, x1, x2, x3 = val[lo], val[lo + 1], val[lo + 2], val[lo + 3]
                s0, s1, s2 = x0 & x1 & x2, maj(x0, x1, x2), x0 | x1 | x2
                o0, o1, o2, o3 = s0 & x3, maj(s0, s1, x3), maj(s1, s2, x3), s2 | x3
                visit(k, nd, dict(s0=s0, s1=s1, s2=s2, x3=x3, o=(o0, o1, o2, o3)))
                val[lo], val[lo + 1], val[lo + 2], val[lo + 3] = o0, o1, o2, o3
            else:
                X, Y = val[nd["a"]], val[nd["b"]]
                l, h = X & Y, X | Y
                visit(k, nd, dict(X=X, Y=Y, lo=l, hi=h))
                val[nd["a"]], val[nd["b"]] = l, h
        return val


def producer_inputs(nd, rec, out):
    """(X, Y) such that the producer's output `out` is min(X, Y) ('lo') or max(X, Y) ('hi')"""
    if nd["kind"] == "CE":
        return rec["X"], rec["Y"]
    return (rec["s0"], rec["x3"]) if out == "lo" else (rec["s2"], rec["x3"])


def candidate_edges(net, rng):
    """all (producer node j, output o, consumer CE k, side) with the rewrite condition checked on every level context"""
    n = net.n
    cand = {}   # (j, o, k, side) -> still valid
    levels = [s for s in (8, 16, 32, 64, 128, 256) if s <= n]
    def batches():
        for s in levels:
            for st in level_state_chunks(n, s, rng):
                yield s, st
        yield 0, random_states(n, 4096, rng)

    for size, st in batches():
        recs = {}
        uses = {}  # producer -> number of consumers still to come (so the vectors can be freed)

        def visit(k, nd, rec):
            recs[k] = rec
            if nd["kind"] != "CE":
                return
            for side, p in (("a", nd["pa"]), ("b", nd["pb"])):
                if p is None:
                    continue
                j, o = p
                # a level's states are exhaustive for the consumers of that level only (an edge lies inside the consumer's block)
                key = (j, o, k, side)
                u = rec["Y"] if side == "a" else rec["X"]
                Xj, Yj = producer_inputs(net.nodes[j], recs[j], o)
                if o == "lo":
                    bad = u & ~(Xj | Yj)      # u = 1 while max(X, Y) = 0  <=>  u > max(X, Y)
                else:
                    bad = ~u & (Xj & Yj)      # u = 0 while min(X, Y) = 1  <=>  u < min(X, Y)
                ok = not bad.any()
                cand[key] = cand.get(key, True) and ok
            # free the records of producers both of whose outputs have been consumed
            for p in (nd["pa"], nd["pb"]):
                if p is not None:
                    uses[p[0]] = uses.get(p[0], 0) + 1
                    full = 2
                    if uses[p[0]] >= full and p[0] in recs:
                        del recs[p[0]]

        final = net.simulate(st, visit)
        for w in range(n - 1):  # the plain network sorts every state of the batch
            assert not (final[w] & ~final[w + 1]).any()
    return [key for key, ok in cand.items() if ok]


def choose(net, edges):
    """0/1 programme: f_k = exchange k is a fused consumer, e = this producer output is absorbed into that consumer.
         sum of e into k  = f_k          (a fused consumer absorbs exactly one of its inputs)
         e(j -> k) + f_j <= 1            (an absorbed value must be a plain two-input min / max)
       maximise the number of absorbed values."""
    from scipy.optimize import Bounds, LinearConstraint, milp
    from scipy.sparse import lil_matrix

    ce = [k for k, nd in enumerate(net.nodes) if nd["kind"] == "CE"]
    col = {k: i for i, k in enumerate(ce)}
    nf, ne = len(ce), len(edges)
    A = lil_matrix((nf + ne, nf + ne))
    lb = np.zeros(nf + ne)
    ub = np.zeros(nf + ne)
    into = {}
    for ei, (j, o, k, side) in enumerate(edges):
        into.setdefault(k, []).append(ei)
    r = 0
    for k in ce:
        for ei in into.get(k, []):
            A[r, nf + ei] = 1
        A[r, col[k]] = -1
        r += 1
    for ei, (j, o, k, side) in enumerate(edges):
        if net.nodes[j]["kind"] == "CE":
            A[r, nf + ei] = 1
            A[r, col[j]] = 1
            lb[r], ub[r] = -np.inf, 1
            r += 1
    c = np.zeros(nf + ne)
    c[:nf] = -1.0
    res = milp(c, constraints=LinearConstraint(A.tocsr()[:r], lb[:r], ub[:r]), integrality=np.ones(nf + ne), bounds=Bounds(0, 1),
               options=dict(time_limit=120.0, mip_rel_gap=0.0 if nf < 2000 else 0.002))
    assert res.x is not None, res.message
    x = np.round(res.x).astype(int)
    return [edges[ei] for ei in range(ne) if x[nf + ei]]


# ---------------------------------------------------------------------------------------------------------------------------
# the rewritten programme: a list of SSA operations  (dst, op, srcs), dst / srcs are value names
# ---------------------------------------------------------------------------------------------------------------------------
def build_program(net, chosen):
    """returns (ops, out): ops = [(dst, op, (src, ...))], out[w] = value name on wire w at the end.  Values are 'i<w>' (inputs) or
    't<k>'.  op in min2 max2 min3 med3 max3 ce (dst = (lo, hi): both outputs of a plain exchange, the includer picks the form)."""
    absorbed = {(j, o): (k, side) for (j, o, k, side) in chosen}
    fused = {k: (j, o, side) for (j, o, k, side) in chosen}
    ops = []
    cnt = [0]

    def new():
        cnt[0] += 1
        return f"t{cnt[0]}"

    cur = {w: f"i{w}" for w in range(net.n)}
    pin = {}   # producer node -> (Xname, Yname) for each output
    quarter = max(net.n // 4, 4)
    for k, nd in enumerate(net.nodes):
        if nd["kind"] == "S4":
            lo = nd["lo"]
            if lo % quarter == 0:   # nothing before this point reads an input of quarter lo / quarter (halves-first order)
                ops.append((None, "hook", (lo // quarter,)))
            x0, x1, x2, x3 = (cur[lo + i] for i in range(4))
            s0, s1, s2 = new(), new(), new()
            ops.append((s0, "min3", (x0, x1, x2)))
            ops.append((s1, "med3", (x0, x1, x2)))
            ops.append((s2, "max3", (x0, x1, x2)))
            pin[k] = {"lo": (s0, x3), "hi": (s2, x3)}
            o1, o2 = new(), new()
            if (k, "lo") in absorbed:
                o0 = None
            else:
                o0 = new()
                ops.append((o0, "min2", (s0, x3)))
            ops.append((o1, "med3", (s0, s1, x3)))
            ops.append((o2, "med3", (s1, s2, x3)))
            if (k, "hi") in absorbed:
                o3 = None
            else:
                o3 = new()
                ops.append((o3, "max2", (s2, x3)))
            cur[lo], cur[lo + 1], cur[lo + 2], cur[lo + 3] = o0, o1, o2, o3
            continue
        a, b = nd["a"], nd["b"]
        if k in fused:
            j, o, side = fused[k]
            Xj, Yj = pin[j][o]
            u = cur[b] if side == "a" else cur[a]
            assert u is not None and (cur[a] if side == "a" else cur[b]) is None
            l, h = new(), new()
            if o == "lo":   # w = min(X, Y):  min(u, w) = min3(u, X, Y),  max(u, w) = med3(u, X, Y)
                ops.append((l, "min3", (u, Xj, Yj)))
                ops.append((h, "med3", (u, Xj, Yj)))
            else:           # w = max(X, Y):  min(u, w) = med3(u, X, Y),  max(u, w) = max3(u, X, Y)
                ops.append((l, "med3", (u, Xj, Yj)))
                ops.append((h, "max3", (u, Xj, Yj)))
            cur[a], cur[b] = l, h
            pin[k] = None
            continue
        X, Y = cur[a], cur[b]
        assert X is not None and Y is not None, (k, nd)
        pin[k] = {"lo": (X, Y), "hi": (X, Y)}
        keep_lo, keep_hi = (k, "lo") not in absorbed, (k, "hi") not in absorbed
        l = new() if keep_lo else None
        h = new() if keep_hi else None
        if keep_lo and keep_hi:
            ops.append(((l, h), "ce", (X, Y)))
        elif keep_lo:
            ops.append((l, "min2", (X, Y)))
        elif keep_hi:
            ops.append((h, "max2", (X, Y)))
        cur[a], cur[b] = l, h
    assert all(cur[w] is not None for w in range(net.n))
    return ops, [cur[w] for w in range(net.n)]


def count_ops(ops):
    return sum(2 if op == "ce" else (0 if op == "hook" else 1) for _, op, _ in ops)


def verify_program(net, ops, out, rng):
    """plain network and programme side by side on every level context + random vectors: the programme's outputs are the plain
    network's (sorted) outputs, on every state"""
    n = net.n
    levels = [s for s in (8, 16, 32, 64, 128, 256) if s <= n]
    last_use = {}
    ops = [o for o in ops if o[1] != "hook"]
    for idx, (dst, op, srcs) in enumerate(ops):
        for s in srcs:
            last_use[s] = idx
    for name in out:
        last_use[name] = len(ops)
    def batches():
        for s in levels:
            yield from level_state_chunks(n, s, rng)
        yield random_states(n, 8192, rng)

    for st in batches():
        plain = net.simulate(st, lambda *a: None)
        val = {f"i{w}": st[w] for w in range(n)}
        for idx, (dst, op, srcs) in enumerate(ops):
            v = [val[s] for s in srcs]
            if op == "ce":
                val[dst[0]], val[dst[1]] = v[0] & v[1], v[0] | v[1]
            elif op == "min2":
                val[dst] = v[0] & v[1]
            elif op == "max2":
                val[dst] = v[0] | v[1]
            elif op == "min3":
                val[dst] = v[0] & v[1] & v[2]
            elif op == "max3":
                val[dst] = v[0] | v[1] | v[2]
            elif op == "med3":
                val[dst] = maj(*v)
            for s in srcs:
                if last_use.get(s) == idx and s in val:
                    del val[s]
        for w in range(n):
            assert np.array_equal(val[out[w]], plain[w]), (n, w)
        for w in range(n - 1):
            assert not (plain[w] & ~plain[w + 1]).any()


CHOICE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sortnet_choice.json")
SOLVE = "--solve" in sys.argv   # ignore the stored choices: find the candidates and solve the 0/1 programmes again


def stored_choice(n):
    import json
    if SOLVE or not os.path.exists(CHOICE_PATH):
        return None
    c = json.load(open(CHOICE_PATH)).get(str(n))
    return None if c is None else (c["candidates"], [tuple(e) for e in c["chosen"]])


def store_choice(n, candidates, chosen):
    import json
    d = json.load(open(CHOICE_PATH)) if os.path.exists(CHOICE_PATH) else {}
    d[str(n)] = dict(candidates=candidates, chosen=[list(e) for e in chosen])
    with open(CHOICE_PATH, "w") as f:
        json.dump(d, f, separators=(",", ":"))
        f.write("\n")


def fused_network(n, seed=1):
    """The rewritten programme of the n-wire network.  WHICH exchanges absorb which values (the 0/1 programme's answer: for 256
    wires a two-minute solve under a time limit, so not reproducible to the bit) is kept in tools/sortnet_choice.json; the header
    is a deterministic function of that file, and whatever the file says is verified here against the plain network on every
    0-1 state of every merge level before a line is emitted -- a wrong or stale choice cannot produce a header."""
    rng = np.random.default_rng(seed)
    net = Net(n)
    stored = stored_choice(n)
    if stored is None:
        edges = candidate_edges(net, rng)
        chosen = choose(net, edges)
        store_choice(n, len(edges), chosen)
        n_cand = len(edges)
    else:
        n_cand, chosen = stored
    ops, out = build_program(net, chosen)
    verify_program(net, ops, out, np.random.default_rng(seed + 1))
    return ops, out, dict(exchanges=sum(1 for nd in net.nodes if nd["kind"] == "CE"), base=sum(1 for nd in net.nodes if nd["kind"] == "S4"),
                          candidates=n_cand, absorbed=len(chosen), instructions=count_ops(ops))


# ---------------------------------------------------------------------------------------------------------------------------
# rendering
# ---------------------------------------------------------------------------------------------------------------------------
def render_plain(n, out):
    ces = batcher(n)
    check(n, ces)
    out.append(f"template <> struct SortNet<{n}> {{  // {len(ces)} compare-exchanges")
    out.append(f"    static constexpr int kCE = {len(ces)};")
    out.append("    template <typename T> static __device__ __forceinline__ void sort(T (&v)[%d]) {" % n)
    line = "        "
    toks, i = [], 0
    while i < len(ces):  # the base case: 5 compare-exchanges on four neighbouring wires -> one AB_SORT4
        a = ces[i][0]
        if n >= 4 and a % 4 == 0 and ces[i:i + 5] == [(a, a + 1), (a + 2, a + 3), (a, a + 2), (a + 1, a + 3), (a + 1, a + 2)]:
            toks.append(f"AB_SORT4({a},{a + 1},{a + 2},{a + 3}) ")
            i += 5
        else:
            toks.append(f"AB_CE({ces[i][0]},{ces[i][1]}) ")
            i += 1
    for tok in toks:
        if len(line) + len(tok) > 118:
            out.append(line.rstrip())
            line = "        "
        line += tok
    out.append(line.rstrip())
    out.append("    }")


def render_fused(n, out, stats):
    ops, outs, st = fused_network(n)
    stats[n] = st
    out.append(f"    // rewritten over min3 / med3 / max3: {st['exchanges']} exchanges + {st['base']} base cases, {st['absorbed']} values absorbed"
               f" -> {st['instructions']} instructions")
    out.append(f"    static constexpr int kFusedInstructions = {st['instructions']};")
    out.append("    // hook(std::integral_constant<int, q>): called before the first operation that reads an input of quarter q (inputs")
    out.append("    // %d q .. %d q + %d) -- the caller may still be waiting for / patching those inputs up to that point" % (max(n // 4, 4), max(n // 4, 4), max(n // 4, 4) - 1))
    out.append("    template <typename Hook> static __device__ __forceinline__ void sort_fused(float (&v)[%d], Hook &&hook) {" % n)

    def nm(s):
        return f"v[{s[1:]}]" if s[0] == "i" else s

    line = "        "
    for dst, op, srcs in ops:
        a = "" if op == "hook" else ", ".join(nm(s) for s in srcs)
        if op == "hook":
            tok = f"hook(std::integral_constant<int, {srcs[0]}>{{}}); "
        elif op == "ce":
            tok = f"float {dst[0]}, {dst[1]}; AB_SN_CE({dst[0]}, {dst[1]}, {a}); "
        else:
            tok = f"const float {dst} = AB_SN_{op.upper()}({a}); "
        if len(line) + len(tok) > 150:
            out.append(line.rstrip())
            line = "        "
        line += tok
    if line.strip():
        out.append(line.rstrip())
    line = "        "
    for w, s in enumerate(outs):
        tok = f"v[{w}] = {nm(s)}; "
        if len(line) + len(tok) > 150:
            out.append(line.rstrip())
            line = "        "
        line += tok
    out.append(line.rstrip())
    out.append("    }")
    out.append("    static __device__ __forceinline__ void sort_fused(float (&v)[%d]) { sort_fused(v, [](auto) {}); }" % n)
    if n in PADDED_SIZES:
        # The SAME programme on values that are either a float or the tag snpad::Inf.  A stack of fewer than %d frames pads the top
        # wires with +inf; Batcher's network is standard (every exchange leaves the minimum on the lower wire), so a pad never
        # leaves its wire and every operation that touches one collapses AT COMPILE TIME: min(x, inf) = x, max(x, inf) = inf,
        # med3(x, y, inf) = max(x, y), an exchange with a pad is no instruction at all.  200 frames then run the part of the
        # 256-wire network that 208 wires need.  Value for value the full programme with real +inf inputs (each snpad overload
        # returns what the instruction would return), so everything the generator proved about it carries over.
        out.append("    // the same programme with wires >= NREAL known to be +inf pads (snpad: operations on pads vanish at compile time)")
        out.append("    template <int NREAL, typename Hook> static __device__ __forceinline__ void sort_fused_n(float (&v)[%d], Hook &&hook) {" % n)

        def pn(s):
            return f"snpad::in<{s[1:]}, NREAL>(v)" if s[0] == "i" else s

        line = "        "
        for dst, op, srcs in ops:
            a = "" if op == "hook" else ", ".join(pn(s) for s in srcs)
            if op == "hook":
                tok = f"hook(std::integral_constant<int, {srcs[0]}>{{}}); "
            elif op == "ce":
                tok = f"const auto p_{dst[0]} = snpad::ce({a}); const auto {dst[0]} = p_{dst[0]}.lo; const auto {dst[1]} = p_{dst[0]}.hi; "
            else:
                tok = f"const auto {dst} = snpad::{op}({a}); "
            if len(line) + len(tok) > 150:
                out.append(line.rstrip())
                line = "        "
            line += tok
        if line.strip():
            out.append(line.rstrip())
        line = "        "
        for w, s in enumerate(outs):
            tok = f"snpad::out<{w}>(v, {pn(s)}); "
            if len(line) + len(tok) > 150:
                out.append(line.rstrip())
                line = "        "
            line += tok
        out.append(line.rstrip())
        out.append("    }")


def render(stats=None):
    stats = {} if stats is None else stats
    body = []
    for n in SIZES:
        render_plain(n, body)
        if n in FUSED_SIZES:
            render_fused(n, body, stats)
        body.append("};")
    out = []
    out.append("// GENERATED by tools/gen_sortnet.py -- do not edit.  Batcher odd-even merge-sort networks (halves first), as written and")
    out.append("// rewritten over the three-input min3 / med3 / max3 (see the generator for the rewrite and how every network is checked).")
    out.append("#pragma once")
    out.append("#include <type_traits>")
    out.append("// plain form -- AB_CE(a, b): compare-exchange so that v[a] <= v[b] afterwards (defined by the includer).")
    out.append("// AB_SORT4(a, b, c, d): sorts four UNSORTED wires (the base case of every network here).  The includer may define it with")
    out.append("// three-input operations (min3 / med3 / max3 + an insertion: 7 instructions); the default is Batcher's 5 compare-exchanges.")
    out.append("#ifndef AB_SORT4")
    out.append("#define AB_SORT4(a, b, c, d) AB_CE(a, b) AB_CE(c, d) AB_CE(a, c) AB_CE(b, d) AB_CE(b, c)")
    out.append("#endif")
    out.append("// rewritten form (sort_fused, float only) -- the includer may define AB_SN_MIN2 / MAX2 / MIN3 / MED3 / MAX3 (value expressions)")
    out.append("// and AB_SN_CE(lo, hi, x, y) (both outputs of a plain exchange); the defaults are the obvious ones.  No NaN may reach a network.")
    for name, expr in (("MIN2(a, b)", "fminf(a, b)"), ("MAX2(a, b)", "fmaxf(a, b)"), ("MIN3(a, b, c)", "fminf(fminf(a, b), c)"),
                       ("MAX3(a, b, c)", "fmaxf(fmaxf(a, b), c)"), ("MED3(a, b, c)", "__builtin_amdgcn_fmed3f(a, b, c)")):
        out.append(f"#ifndef AB_SN_{name.split('(')[0]}")
        out.append(f"#define AB_SN_{name} {expr}")
        out.append("#endif")
    out.append("#ifndef AB_SN_CE")
    out.append("#define AB_SN_CE(lo, hi, x, y) { lo = AB_SN_MIN2(x, y); hi = AB_SN_MAX2(x, y); }")
    out.append("#endif")
    out.append("// sort_fused_n<NREAL>: values are floats or the tag Inf (a wire >= NREAL: a +inf pad that never leaves its wire).  Every overload")
    out.append("// returns what the instruction would return on a real +inf -- as a type where the answer is a pad, with fewer instructions where it is not.")
    out.append("namespace snpad {")
    out.append("struct Inf {};")
    out.append("template <class T> constexpr bool pad = std::is_same<T, Inf>::value;")
    out.append("template <class A, class B> struct Pair { A lo; B hi; };")
    out.append("template <int I, int NREAL, int N> __device__ __forceinline__ auto in(const float (&v)[N]) { if constexpr (I < NREAL) return v[I]; else return Inf{}; }")
    out.append("template <int I, int N, class T> __device__ __forceinline__ void out(float (&v)[N], T t) { if constexpr (!pad<T>) v[I] = t; }  // (a pad wire holds +inf already)")
    out.append("template <class A, class B> __device__ __forceinline__ auto min2(A a, B b) {")
    out.append("    if constexpr (pad<A>) return b; else if constexpr (pad<B>) return a; else return AB_SN_MIN2(a, b);")
    out.append("}")
    out.append("template <class A, class B> __device__ __forceinline__ auto max2(A a, B b) {")
    out.append("    if constexpr (pad<A> || pad<B>) return Inf{}; else return AB_SN_MAX2(a, b);")
    out.append("}")
    out.append("template <class A, class B, class C> __device__ __forceinline__ auto min3(A a, B b, C c) {")
    out.append("    if constexpr (pad<A>) return min2(b, c); else if constexpr (pad<B>) return min2(a, c); else if constexpr (pad<C>) return min2(a, b);")
    out.append("    else return AB_SN_MIN3(a, b, c);")
    out.append("}")
    out.append("template <class A, class B, class C> __device__ __forceinline__ auto max3(A a, B b, C c) {")
    out.append("    if constexpr (pad<A> || pad<B> || pad<C>) return Inf{}; else return AB_SN_MAX3(a, b, c);")
    out.append("}")
    out.append("template <class A, class B, class C> __device__ __forceinline__ auto med3(A a, B b, C c) {  // the median of {x, y, +inf} is max(x, y)")
    out.append("    if constexpr (pad<A>) return max2(b, c); else if constexpr (pad<B>) return max2(a, c); else if constexpr (pad<C>) return max2(a, b);")
    out.append("    else return AB_SN_MED3(a, b, c);")
    out.append("}")
    out.append("template <class A, class B> __device__ __forceinline__ auto ce(A x, B y) {")
    out.append("    if constexpr (pad<A> && pad<B>) return Pair<Inf, Inf>{};")
    out.append("    else if constexpr (pad<A>) return Pair<B, Inf>{y, Inf{}};")
    out.append("    else if constexpr (pad<B>) return Pair<A, Inf>{x, Inf{}};")
    out.append("    else { float lo, hi; AB_SN_CE(lo, hi, x, y); return Pair<float, float>{lo, hi}; }")
    out.append("}")
    out.append("}  // namespace snpad")
    out.append("template <int NP> struct SortNet;")
    out.extend(body)
    return "\n".join(out) + "\n"


PATH = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "astroburst_amd", "csrc", "sortnet_gen.hpp"))


def main():
    if "--count" in sys.argv:  # instruction counts only (no header)
        for n in FUSED_SIZES:
            if n > int(sys.argv[sys.argv.index("--count") + 1]):
                break
            print(n, fused_network(n)[2])
        return
    stats = {}
    text = render(stats)
    if "--check" in sys.argv:
        if open(PATH).read() != text:
            print("sortnet_gen.hpp is stale: run python tools/gen_sortnet.py")
            sys.exit(1)
        print("sortnet_gen.hpp is current")
        return
    with open(PATH, "w") as f:
        f.write(text)
    print("wrote", PATH)
    for n, st in stats.items():
        print(f"  {n:4d}: {st}")


if __name__ == "__main__":
    main()
