"""GPU parity (run on the MI355X box: pytest -m gpu).  Everything goes through the C ABI
(libbowtie_amd.so -> HIP kernels); the oracle and the reference's golden outputs are the checkers."""
import json
import os

import numpy as np
import pytest

import common as T
from bowtie_amd import _abi as A
from bowtie_amd import aligner as AL
from bowtie_amd.reads import Read, pack_reads
from bowtie_amd.synth import synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gidx():
    return {n: AL.Index(os.path.join(T.G, n)) for n in ("e_coli", "multi")}


def aligner(gidx, name, kw):
    return AL.Aligner(gidx[name], A.make_policy(**kw))


@pytest.mark.parametrize("index", ["e_coli", "multi"])
def test_probe_rank_blocks_equal_the_side_layout(index, gidx):
    """The 32-byte rank blocks the loader derives on the GPU (bt_blk_build_kernel) against the index files' side layout
    they are derived from: every row around the '$' row, around block and side boundaries and at both ends, and
    20 000 random rows, text index and mirror index."""
    al = aligner(gidx, index, T.MODES["v0"])
    info = gidx[index].info
    n, z = int(info.len), int(info.z_off)
    rng = np.random.default_rng(5)
    special = [0, 1, 2, 63, 64, 65, 223, 224, 225, 447, 448, 449, n - 1, n, n + 1] + list(range(max(0, z - 70), min(n + 1, z + 70)))
    rows = np.array(sorted(set(r for r in special if 0 <= r <= n + 1)) + list(rng.integers(0, n + 2, size=20000)), dtype=np.uint32)
    for mirror in (False, True):
        lf_b, L_b = al.probe_rank(rows, mirror=mirror)
        lf_s, L_s = al.probe_rank(rows, mirror=mirror, sides=True)
        assert (lf_b == lf_s).all(), (index, mirror, rows[(lf_b != lf_s).any(axis=1)][:5])
        ok = rows <= n                                 # the BWT has rows 0..len
        assert (L_b[ok] == L_s[ok]).all(), (index, mirror)


@pytest.mark.parametrize("index", ["e_coli", "multi"])
def test_locus_image_equals_the_host_build(index, gidx):
    """The locus image the loader derives on the GPU (bt_loc_sa_kernel / bt_loc_ctx_kernel: dense suffix array + 48 characters
    of left context per row, the reversed 2-bit text, the table of walk lengths) against the host build of the same
    (bt_loc_build_host, one sequential pass over the text), text index and mirror index, every row."""
    import ctypes as C
    import emu_lib as E
    al = aligner(gidx, index, T.MODES["n2"])               # creating a phase-program context derives the image
    L = AL.lib()
    assert L.bt_ctx_get_locus(al._h) == 1 and L.bt_index_locus_bytes(gidx[index]._h) > 0
    e = E.EmuAligner(os.path.join(T.G, index))
    EL = E.lib()
    EL.emu_locus_arrays.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(C.c_uint32)]
    for mirror in (0, 1):
        pl, pt, pw, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        assert EL.emu_locus_arrays(e.h, mirror, C.byref(pl), C.byref(pt), C.byref(pw), C.byref(n)) == 0
        rows, words = n.value + 1, (n.value + 15) // 16
        want_loc = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_uint32)), shape=(rows, 4))
        want_txt = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_uint32)), shape=(words,))
        want_walk = np.ctypeslib.as_array(C.cast(pw, C.POINTER(C.c_uint16)), shape=(rows,))
        loc = np.zeros((rows, 4), dtype=np.uint32); txt = np.zeros(words, dtype=np.uint32); walk = np.zeros(rows, dtype=np.uint16)
        assert L.bt_index_locus_copy(gidx[index]._h, mirror, loc.ctypes.data, txt.ctypes.data, walk.ctypes.data) == 0
        assert (txt == want_txt).all(), (index, mirror, "text")
        assert (walk == want_walk).all(), (index, mirror, "walk lengths")
        assert (loc == want_loc).all(), (index, mirror, "records", np.nonzero((loc != want_loc).any(axis=1))[0][:5])


@pytest.mark.parametrize("mode", ["v0", "v2", "n2", "n3", "n2_k3", "n1_a_m20", "n2_nomaq"])
def test_locus_mode_on_and_off_agree(mode, gidx):
    """One context, the same ragged reads with locus mode and in row space: the same hits, the same op counts (what the
    text decided is tallied as the reference's steps), and far fewer lock-step rounds."""
    kw = T.MODES[mode]
    text = T.joined_text("multi")
    rng = np.random.default_rng(77)
    reads = []
    for i in range(1200):
        ln = int(rng.integers(4, 105))
        b = synth_reads(text, 1, ln, mm_dist=(0, 1, 2, 3), seed=21000 + i, n_frac=0.1, lowq_frac=0.1)
        reads.append(Read(("q%d" % i).encode(), b.seq[0, :ln].copy(), b.qual[0, :ln].tobytes()))
    batch = pack_reads(reads)
    al = aligner(gidx, "multi", kw)
    L = AL.lib()
    out = {}
    for on in (1, 0):
        assert L.bt_ctx_set_locus(al._h, on) == 0 and L.bt_ctx_get_locus(al._h) == on
        c = A.OpCounts()
        out[on] = (al.align(batch, hit_cap=T.hit_cap_for(kw), counts=c), c)
    T.compare_results(out[1][0], out[0][0], mode + " locus mode against row space")
    T.check_op_counts(out[0][1], out[1][1], mode)
    assert out[0][1].loc_records == 0 and out[1][1].loc_records > 0
    assert out[1][1].lane_iters < out[0][1].lane_iters


def test_probe_rank_known_answers(gidx):
    with open(os.path.join(T.G, "rank_vectors.json")) as f:
        v = json.load(f)
    al = aligner(gidx, "e_coli", T.MODES["v0"])
    rows = np.array([int(r) for r in v["rows"]], dtype=np.uint32)
    lf, L = al.probe_rank(rows)
    for i, r in enumerate(rows):
        e = v["rows"][str(int(r))]
        assert list(lf[i]) == e["lf"], r
        if int(r) != v["zOff"]:
            assert int(L[i]) == e["L"], r
    j, t, o = al.probe_chase(np.array([v["chase"]["row"]], dtype=np.uint32), 36)
    assert (int(j[0]), int(t[0]), int(o[0])) == (v["chase"]["joined"], v["chase"]["tidx"], v["chase"]["toff"])


@pytest.mark.parametrize("name", ["e_coli", "multi"])
def test_probe_rank_and_chase_vs_oracle(gidx, name):
    oi = T.oracle_index(name)
    rng = np.random.default_rng(5)
    rows = np.concatenate([rng.integers(0, oi.fw.len + 1, size=20000),
                           [0, 1, 223, 224, 447, 448, oi.fw.zOff, oi.fw.zOff + 1, oi.fw.len]]).astype(np.uint32)
    al = aligner(gidx, name, T.MODES["n2"])
    for mirror in (False, True):
        lf, L = al.probe_rank(rows, mirror)
        z = oi.ix(mirror).zOff
        for i in range(0, len(rows), 7):
            olf, oL = oi.rank4(int(rows[i]), mirror)
            assert list(lf[i]) == olf
            if rows[i] != z:
                assert int(L[i]) == oL
        sub = rows[:3000]
        j, t, o = al.probe_chase(sub, 30, mirror)
        for i in range(0, len(sub), 3):
            off, _ = oi.chase(int(sub[i]), mirror)
            assert int(j[i]) == off
            assert (int(t[i]), int(o[i])) == oi.joined_to_text(30, off, mirror)


@pytest.mark.parametrize("run", T.golden_runs(), ids=lambda r: r["file"][:-7])
def test_gpu_matches_reference_sam(run, gidx):
    """HIP path -> SAM == the unmodified reference's SAM, byte for byte."""
    batch = T.read_set(run["index"], run["reads"])
    kw = T.MODES[run["mode"]]
    al = aligner(gidx, run["index"], kw)
    res = al.align(batch, hit_cap=T.hit_cap_for(kw))
    T.check_against_golden(run, res, batch, gidx[run["index"]].refnames)


@pytest.mark.parametrize("mode", ["v0", "v1", "v2", "n0", "n1", "n2", "n3", "n2_k3", "n2_nomaq", "n1_a_m20"])
def test_gpu_vs_oracle_ragged(mode, gidx):
    kw = T.MODES[mode]
    text = T.joined_text("multi")
    rng = np.random.default_rng(99)
    reads = []
    for i in range(1500):
        L = int(rng.integers(4, 151))
        b = synth_reads(text, 1, L, mm_dist=(0, 1, 2, 3), seed=1000 + i, n_frac=0.2, lowq_frac=0.1)
        reads.append(Read(("q%d" % i).encode(), b.seq[0, :L].copy(), b.qual[0, :L].tobytes()))
    batch = pack_reads(reads)
    import oracle_lib as OL
    oc, gc = OL.OpCounts(), A.OpCounts()
    want = T.oracle_results("multi", batch, kw, cap=T.hit_cap_for(kw), counts=oc)
    got = aligner(gidx, "multi", kw).align(batch, hit_cap=T.hit_cap_for(kw), counts=gc)
    T.compare_results(got, want, mode)
    T.check_op_counts(oc, gc)


@pytest.mark.parametrize("chars", [9, 11, 13])
def test_gpu_jump_table_results_and_op_counts(chars, monkeypatch):
    """The jump table (bt_rank.h: the range behind a search's first 14 characters from one look-up where the phase may not
    revisit them, instead of ftab's 10 and the steps behind it): indexes loaded with tables of 9, 11 and 13 characters
    (BT_JUMP_CHARS; the test indexes' own ftabs have 7 and 6, and `multi` is too small to get a table by itself), ragged reads with Ns and low qualities and e_coli's 100-bp reads
    in every phase-program mode -- the oracle's results, the oracle's op counts (the steps behind a look-up are tallied as the
    reference takes them), look-ups actually made, and a context with BT_JUMP=0 beside it for the same answers."""
    import ctypes as C
    import oracle_lib as OL
    monkeypatch.setenv("BT_JUMP_CHARS", str(chars))
    idx = {n: AL.Index(os.path.join(T.G, n)) for n in ("e_coli", "multi")}
    L = AL.lib()
    L.bt_index_jump_bytes.restype = C.c_uint64
    assert L.bt_index_jump_bytes(idx["multi"]._h) == 2 * 10 * 4 ** chars
    text = T.joined_text("multi")
    rng = np.random.default_rng(7)
    reads = []
    for i in range(1200):
        ln = int(rng.integers(4, 113))
        b = synth_reads(text, 1, ln, mm_dist=(0, 1, 2, 3), seed=5000 + i, n_frac=0.1, lowq_frac=0.1)
        reads.append(Read(("j%d" % i).encode(), b.seq[0, :ln].copy(), b.qual[0, :ln].tobytes()))
    ragged = pack_reads(reads)
    eco = synth_reads(T.joined_text("e_coli"), 6000, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=77)
    for mode in ("v0", "v1", "v2", "n2", "n3", "n2_k3", "n1_a_m20"):
        kw = T.MODES[mode]
        for name, batch in (("multi", ragged), ("e_coli", eco)):
            oc, gc = OL.OpCounts(), A.OpCounts()
            want = T.oracle_results(name, batch, kw, cap=T.hit_cap_for(kw), counts=oc)
            al = AL.Aligner(idx[name], A.make_policy(**kw))
            got = al.align(batch, hit_cap=T.hit_cap_for(kw), counts=gc)
            T.compare_results(got, want, "jump table %d %s %s" % (chars, name, mode))
            T.check_op_counts(oc, gc, "jump table %d %s %s" % (chars, name, mode))
            lk, st = C.c_uint64(), C.c_uint64()
            L.bt_ctx_jump_counts(al._h, C.byref(lk), C.byref(st))
            assert lk.value > 0, (chars, name, mode)
            if mode == "n2":
                monkeypatch.setenv("BT_JUMP", "0")
                off = AL.Aligner(idx[name], A.make_policy(**kw))
                monkeypatch.delenv("BT_JUMP")
                g0 = A.OpCounts()
                T.compare_results(off.align(batch, hit_cap=T.hit_cap_for(kw), counts=g0), want, "no jump table %s" % name)
                L.bt_ctx_jump_counts(off._h, C.byref(lk), C.byref(st))
                assert lk.value == 0 and g0.lf2 == gc.lf2 and g0.lf1 == gc.lf1 and g0.ftab == gc.ftab and g0.lane_iters > gc.lane_iters


@pytest.mark.parametrize("mode,length,n", [("v0", 36, 20000), ("v2", 76, 20000), ("n2", 100, 20000)])
def test_gpu_vs_oracle_e_coli_synthetic(mode, length, n, gidx):
    kw = T.MODES[mode]
    batch = synth_reads(T.joined_text("e_coli"), n, length, seed=4242 + length)
    want = T.oracle_results("e_coli", batch, kw)
    got = aligner(gidx, "e_coli", kw).align(batch)
    T.compare_results(got, want, mode)


def test_gpu_large_batch_properties(gidx):
    """BASELINE config-2 size (1M x 36 bp, -v 0): properties that need no oracle run --
    idempotence, batch-split invariance, permutation equivariance, and every reported hit
    re-verified against the reference text."""
    text = T.joined_text("e_coli")
    n = 1_000_000
    batch = synth_reads(text, n, 36, mm_dist=(0,), seed=777, n_frac=0.0)
    al = aligner(gidx, "e_coli", T.MODES["v0"])
    r1 = al.align(batch)
    assert T.result_digest(r1) == T.result_digest(al.align(batch))
    aligned = sum(1 for h, _, _ in r1 if h)
    assert aligned == n                                   # exact reads drawn from the text
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    for i in range(0, n, 997):
        h = r1[i][0][0]
        s = batch.seq[i, :36]
        if not h.fw:
            s = comp[s[::-1]]
        assert (text[h.toff:h.toff + 36] == s).all()
    # permutation equivariance on a slice
    from bowtie_amd.reads import ReadBatch
    sl = slice(0, 50000)
    perm = np.random.default_rng(3).permutation(50000)
    pb = ReadBatch(batch.seq[sl][perm], batch.qual[sl][perm], batch.len[sl][perm], batch.seed[sl][perm],
                   [batch.names[i] for i in perm])
    rp = al.align(pb)
    for k in range(0, 50000, 101):
        assert rp[k] == r1[perm[k]]


def test_gpu_empty_and_edge_batches(gidx):
    al = aligner(gidx, "multi", T.MODES["n2"])
    from bowtie_amd.reads import ReadBatch
    empty = ReadBatch(np.zeros((0, 4), np.uint8), np.zeros((0, 4), np.uint8), np.zeros(0, np.uint16),
                      np.zeros(0, np.uint32), [])
    assert al.align(empty) == []
    reads = [Read(b"a", np.array([0, 1, 2], dtype=np.uint8), b"III"),
             Read(b"c", np.array([4, 4, 4, 1, 2, 3, 0, 1, 2], dtype=np.uint8), b"IIIIIIIII"),
             Read(b"one", np.array([2] * 9, dtype=np.uint8), b"I" * 9)]
    batch = pack_reads(reads)
    got = al.align(batch)
    T.compare_results(got, T.oracle_results("multi", batch, T.MODES["n2"]))
    assert got[0][2] & A.BT_ST_SKIPPED and got[1][2] & A.BT_ST_SKIPPED
    with pytest.raises(AL.BowtieAmdError) as e:
        aligner(gidx, "multi", T.MODES["v2"]).align(batch)
    assert e.value.code == A.BT_ERR_READ_SHORT


def test_gpu_max_length_reads(gidx):
    """1024-bp reads (the reference's FixedBitset<1024> ceiling, hit.h:66)."""
    text = T.joined_text("e_coli")
    batch = synth_reads(text, 64, 1024, mm_dist=(0, 1, 2), seed=31, n_frac=0.0)
    for mode in ("v2", "n2"):
        kw = T.MODES[mode]
        T.compare_results(aligner(gidx, "e_coli", kw).align(batch), T.oracle_results("e_coli", batch, kw), mode)


def test_gpu_heavy_read_offload_is_transparent(gidx, monkeypatch):
    """Reads that run long are parked mid-search and resumed by follow-up launches (two levels);
    with tiny thresholds nearly every read takes that path -- results must not change."""
    monkeypatch.setenv("BT_HEAVY0", "25")
    monkeypatch.setenv("BT_HEAVY1", "90")
    monkeypatch.setenv("BT_HEAVY_MIN_BATCH", "1")
    for index, rname, mode in (("multi", "syn100", "n2"), ("multi", "syn50lowq", "n3"), ("e_coli", "syn76", "v2"),
                               ("multi", "syn76", "n1_a_m20")):
        batch = T.read_set(index, rname)
        kw = T.MODES[mode]
        got = aligner(gidx, index, kw).align(batch, hit_cap=T.hit_cap_for(kw))
        T.compare_results(got, T.oracle_results(index, batch, kw, cap=T.hit_cap_for(kw)), "offload " + mode)


def test_gpu_heavy_first_schedule_is_transparent(gidx, monkeypatch):
    """The lanes pick reads up heaviest-expected-first (ftab-count proxy, counting sort on the GPU);
    only the pick-up order changes, never a result."""
    monkeypatch.setenv("BT_SCHEDULE", "1")
    monkeypatch.setenv("BT_SCHEDULE_MIN_BATCH", "1")
    for index, rname, mode in (("multi", "syn100", "n2"), ("e_coli", "syn36", "v0"), ("multi", "syn12", "n2"),
                               ("multi", "syn50lowq", "v2_a")):
        batch = T.read_set(index, rname)
        kw = T.MODES[mode]
        got = aligner(gidx, index, kw).align(batch, hit_cap=T.hit_cap_for(kw))
        T.compare_results(got, T.oracle_results(index, batch, kw, cap=T.hit_cap_for(kw)), "schedule " + mode)


def test_gpu_scratch_overflow_is_retried(gidx, monkeypatch):
    """A read whose backtracking outgrows its per-lane arenas is flagged by the kernel and re-run by
    bt_align_batch through a worst-case-sized context; with absurdly small arenas most reads take
    that road and the results must still equal the oracle's."""
    monkeypatch.setenv("BT_ENTRY_CAP", "12")      # (24 until round 5: locus mode keeps one range-stack entry for a whole stretch that matches the text)
    monkeypatch.setenv("BT_FRAME_CAP", "3")
    monkeypatch.setenv("BT_PARTIAL_CAP", "4")
    for index, rname, mode in (("multi", "syn100", "n2"), ("multi", "syn50lowq", "n3"), ("e_coli", "syn76", "v2"),
                               ("multi", "syn76", "n1_a_m20"), ("multi", "syn36", "n2_k3")):
        batch = T.read_set(index, rname)
        kw = T.MODES[mode]
        al = aligner(gidx, index, kw)
        got = al.align(batch, hit_cap=T.hit_cap_for(kw))
        assert al.last_retried > batch.n // 20, (mode, al.last_retried)
        T.compare_results(got, T.oracle_results(index, batch, kw, cap=T.hit_cap_for(kw)), "overflow-retry " + mode)


def test_gpu_fresh_context_is_ordered_on_its_own_stream(gidx, monkeypatch):
    """Round 6's finding: a context's cursors were zeroed by null-stream hipMemset calls, which return
    before the fill has run -- and the context's stream, being non-blocking, does not wait for the null stream.  With several
    processes on one GPU the fill could run late: after the first launch, before its mismatch-pool cursor was read (the first
    pass then "handed out no entries" and the second pass's lists were written over the first's: the right hit with another
    read's mismatch list), or under the twin context's first launch (the second pass's lists empty).
    Here the null stream is held busy (tests/emu/gpu_stall.hip) while each context is created and runs its first batch, so
    that anything still enqueued on it runs after the batch: on the parent of the fix this test fails every time
    (scripts/r6/repro_null_stream.py runs it against that library), with the fix nothing of a context depends on the null
    stream any more."""
    import torch
    import emu_lib
    stall = emu_lib.stall_lib()
    monkeypatch.setenv("BT_ENTRY_CAP", "12")
    monkeypatch.setenv("BT_FRAME_CAP", "3")
    monkeypatch.setenv("BT_PARTIAL_CAP", "4")
    keep = [aligner(gidx, ix, T.MODES["n2"]) for ix in ("e_coli", "multi")]      # (the locus images exist: creating a context waits for nothing)
    torch.cuda.synchronize()
    for index, rname, mode in (("multi", "syn100", "n2"), ("e_coli", "syn76", "v2"), ("multi", "syn36", "n2_k3")):
        batch = T.read_set(index, rname)
        kw = T.MODES[mode]
        assert stall.gpu_stall(None, 250) == 0              # the null stream is busy for a quarter of a second ...
        al = aligner(gidx, index, kw)                       # ... while the context is created ...
        keep.append(al)                                     # (destroying one frees device memory, which waits for the device)
        got = al.align(batch, hit_cap=T.hit_cap_for(kw))    # ... and searches its first batch, second pass and all
        assert al.last_retried > batch.n // 20, (mode, al.last_retried)
        T.compare_results(got, T.oracle_results(index, batch, kw, cap=T.hit_cap_for(kw)), "busy null stream " + mode)
        torch.cuda.synchronize()


def test_gpu_hits_verify_against_text_large(gidx):
    """2 M reads through the device-pointer path (what bench.py times), then every reported hit
    re-derived from the text on the GPU (bowtie_amd/verify.py): windows, mismatch lists, policy, cost."""
    import ctypes as C
    import torch
    from bowtie_amd import verify as V
    from bowtie_amd.synth import synth_reads_torch
    dev = torch.device("cuda", 0)
    ln, plen, rstarts = V.read_fragments(os.path.join(T.G, "e_coli"))
    text_t = torch.from_numpy(T.joined_text("e_coli").copy()).to(dev)
    for mode, L in (("n2", 100), ("v2", 76)):
        kw = T.MODES[mode]
        n = 2_000_000
        rb = synth_reads_torch(text_t, n, L, seed=31 + L)
        hits = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
        n_hits = torch.zeros(n, dtype=torch.int32, device=dev)
        status = torch.zeros(n, dtype=torch.uint8, device=dev)
        pool = torch.zeros(n * 8, dtype=torch.int16, device=dev)
        al = aligner(gidx, "e_coli", kw)
        rbc = A.ReadBatchC(n, rb["stride"], rb["seq"].data_ptr(), rb["qual"].data_ptr(), rb["len"].data_ptr(), rb["seed"].data_ptr())
        hbc = A.HitBatchC(1, hits.data_ptr(), n_hits.data_ptr(), status.data_ptr(), pool.data_ptr(), n * 8, 0)
        torch.cuda.synchronize()        # torch filled these arrays on ITS stream: the context's stream does not wait for that one
        assert AL.lib().bt_align_batch_device(al._h, C.byref(rbc), C.byref(hbc), None) == 0
        assert AL.lib().bt_ctx_sync(al._h) == 0
        r = V.verify_hits(text_t, ln, rstarts, rb["seq"], rb["qual"], L, hits, n_hits, pool, kw)
        assert r["checked"] > 0.5 * n
        assert {k: v for k, v in r.items() if k != "checked"} == dict(bad_window=0, bad_mm_count=0, bad_mm_list=0, bad_policy=0, bad_cost=0)


def _device_align(al, batch, stride, hit_cap):
    """bt_align_batch_device on a copy of `batch` in HBM with rows `stride` bytes apart -> unpack_hits()'s list."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    n = batch.n
    seq = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    qual = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    seq[:, :batch.stride] = torch.from_numpy(np.ascontiguousarray(batch.seq)).to(dev)
    qual[:, :batch.stride] = torch.from_numpy(np.ascontiguousarray(batch.qual)).to(dev)
    ln = torch.from_numpy(batch.len.astype(np.int16)).to(dev)
    seed = torch.from_numpy(batch.seed.view(np.int32).copy()).to(dev)
    hits = torch.zeros(n * hit_cap * 24, dtype=torch.uint8, device=dev)
    n_hits = torch.zeros(n, dtype=torch.int32, device=dev)
    status = torch.zeros(n, dtype=torch.uint8, device=dev)
    pool = torch.zeros(n * hit_cap * 8, dtype=torch.int16, device=dev)
    rbc = A.ReadBatchC(n, stride, seq.data_ptr(), qual.data_ptr(), ln.data_ptr(), seed.data_ptr())
    hbc = A.HitBatchC(hit_cap, hits.data_ptr(), n_hits.data_ptr(), status.data_ptr(), pool.data_ptr(), pool.numel(), 0)
    torch.cuda.synchronize()            # torch filled these arrays on ITS stream: the context's stream does not wait for that one
    assert AL.lib().bt_align_batch_device(al._h, C.byref(rbc), C.byref(hbc), None) == 0
    assert AL.lib().bt_ctx_sync(al._h) == 0
    pol = al.policy
    return AL.unpack_hits(n, hit_cap, hits.cpu().numpy().view(A.HIT_DTYPE), n_hits.cpu().numpy().view(np.uint32), status.cpu().numpy(),
                          pool.cpu().numpy().view(np.uint16), int(pol.khits), int(pol.mhits), bool(pol.all_hits),
                          sample_max=bool(pol.sample_max))


@pytest.mark.parametrize("rname", ["syn100", "syn110"])
def test_gpu_device_path_settles_the_build_on_the_device(rname, gidx):
    """Rows 112 bytes apart: whether the three-blocks-per-CU build (reads <= 104) may run depends on the longest
    read, which only the device knows.  Both builds are enqueued, gated on the reduced maximum -- the host does
    not wait -- and whichever runs gives the host path's results."""
    batch = T.read_set("multi", rname)
    kw = T.MODES["n2_k3"]
    al = aligner(gidx, "multi", kw)
    want = al.align(batch, hit_cap=8)
    got = _device_align(al, batch, 112, 8)
    assert b"gated" in AL.lib().bt_ctx_last_kernel_name(al._h)
    T.compare_results(got, want, "device path, stride 112, " + rname)


def test_gpu_device_path_retries_overflowed_reads_on_the_stream(gidx, monkeypatch):
    """bt_align_batch_device with absurdly small arenas: the reads that outgrow them are collected and searched again
    on the same stream (no host copy); what the caller reads back after the sync is complete and equals the oracle's."""
    monkeypatch.setenv("BT_DEVICE_RETRY", "1")                  # the default since round 3
    monkeypatch.setenv("BT_ENTRY_CAP", "12")      # (24 until round 5: locus mode keeps one range-stack entry for a whole stretch that matches the text)
    monkeypatch.setenv("BT_FRAME_CAP", "3")
    monkeypatch.setenv("BT_PARTIAL_CAP", "4")
    for index, rname, mode in (("multi", "syn100", "n2"), ("multi", "syn50lowq", "n3"), ("e_coli", "syn76", "v2"), ("multi", "syn36", "n2_k3"),
                               ("multi", "syn150", "n2")):
        batch = T.read_set(index, rname)
        kw = T.MODES[mode]
        al = aligner(gidx, index, kw)
        cap = T.hit_cap_for(kw)
        got = _device_align(al, batch, (batch.stride + 15) // 16 * 16, cap)
        assert AL.lib().bt_ctx_last_retried(al._h) > batch.n // 20, mode
        assert not any(st & A.BT_ST_OVERFLOW for _, _, st in got)
        T.compare_results(got, T.oracle_results(index, batch, kw, cap=cap), "device retry " + mode)


def _device_align_many(al, batches, stride, hit_cap):
    """A run of bt_align_batch_device calls on one context (each batch with its own arrays in HBM), one bt_ctx_sync at
    the end -> [unpack_hits() list per batch]."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    keep = []
    for batch in batches:
        n = batch.n
        seq = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        qual = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        seq[:, :batch.stride] = torch.from_numpy(np.ascontiguousarray(batch.seq)).to(dev)
        qual[:, :batch.stride] = torch.from_numpy(np.ascontiguousarray(batch.qual)).to(dev)
        ln = torch.from_numpy(batch.len.astype(np.int16)).to(dev)
        seed = torch.from_numpy(batch.seed.view(np.int32).copy()).to(dev)
        hits = torch.zeros(n * hit_cap * 24, dtype=torch.uint8, device=dev)
        n_hits = torch.full((n,), -1, dtype=torch.int32, device=dev)
        status = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        pool = torch.zeros(n * hit_cap * 8, dtype=torch.int16, device=dev)
        rbc = A.ReadBatchC(n, stride, seq.data_ptr(), qual.data_ptr(), ln.data_ptr(), seed.data_ptr())
        hbc = A.HitBatchC(hit_cap, hits.data_ptr(), n_hits.data_ptr(), status.data_ptr(), pool.data_ptr(), pool.numel(), 0)
        keep.append((seq, qual, ln, seed, hits, n_hits, status, pool, rbc, hbc, n))
        torch.cuda.synchronize()        # torch filled these arrays on ITS stream: the context's stream does not wait for that one
        assert AL.lib().bt_align_batch_device(al._h, C.byref(rbc), C.byref(hbc), None) == 0
    assert AL.lib().bt_ctx_sync(al._h) == 0
    pol = al.policy
    out = []
    for (_, _, _, _, hits, n_hits, status, pool, _, _, n) in keep:
        out.append(AL.unpack_hits(n, hit_cap, hits.cpu().numpy().view(A.HIT_DTYPE), n_hits.cpu().numpy().view(np.uint32),
                                  status.cpu().numpy(), pool.cpu().numpy().view(np.uint16), int(pol.khits), int(pol.mhits),
                                  bool(pol.all_hits), sample_max=bool(pol.sample_max)))
    return out


@pytest.mark.parametrize("mode", ["n2", "v2", "n3", "n2_k3", "n1_a_m20"])
def test_gpu_carry_over_between_batches(mode, gidx, monkeypatch):
    """bt_ctx_set_carry: with a two-workgroup grid every batch leaves reads running when its cursor runs dry; they are
    parked, resumed by the next call, and the last ones finished by bt_ctx_sync.  Each batch's results -- written to
    its own arrays, some of them by the launch after its own -- equal the oracle's."""
    monkeypatch.setenv("BT_MAX_BLOCKS", "2")
    kw = T.MODES[mode]
    cap = T.hit_cap_for(kw)
    al = aligner(gidx, "multi", kw)
    assert AL.lib().bt_ctx_set_carry(al._h, 1) == 0
    assert AL.lib().bt_ctx_set_max_read_len(al._h, 100) == 0      # rows are 112 apart; no read is longer than 100
    names = ["syn100", "syn36", "syn50lowq", "syn76", "syn100", "syn12"] if "all_hits" not in kw else ["syn100", "syn36", "syn50lowq", "syn76"]
    batches = [T.read_set("multi", r) for r in names]
    got = _device_align_many(al, batches, 112, cap)
    assert AL.lib().bt_ctx_last_carried(al._h) > 0
    for r, b, g in zip(names, batches, got):
        T.compare_results(g, T.oracle_results("multi", b, kw, cap=cap), "carry-over %s %s" % (mode, r))


@pytest.mark.parametrize("per,age", [(1000, 12), (250, 60)], ids=["40x1000_age12", "160x250_age60"])
def test_gpu_carry_over_many_small_launches_equal_one_big(per, age, gidx, monkeypatch):
    """Size-independent property: 40 k reads as one batch without carry-over and as 40 carried batches of 1 k (512
    lanes: every launch parks what it was doing, long reads ride through several) give the same per-read results.  The
    second case goes round the ring of batches (BT_BATCH_RING = 64, bt_core.h) two and a half times, with reads allowed to ride
    through 60 launches."""
    monkeypatch.setenv("BT_MAX_BLOCKS", "2")
    kw = T.MODES["n2"]
    text = T.joined_text("e_coli")
    big = synth_reads(text, 40000, 76, mm_dist=(0, 1, 2, 2, 3, 4), seed=4242)
    al = aligner(gidx, "e_coli", kw)
    whole = _device_align(al, big, 80, 1)
    al2 = aligner(gidx, "e_coli", kw)
    assert AL.lib().bt_ctx_set_carry(al2._h, age) == 0         # a read may ride along for that many launches
    from bowtie_amd.reads import ReadBatch
    parts = [ReadBatch(big.seq[i:i + per], big.qual[i:i + per], big.len[i:i + per], big.seed[i:i + per], big.names[i:i + per])
             for i in range(0, 40000, per)]
    got = _device_align_many(al2, parts, 80, 1)
    assert AL.lib().bt_ctx_last_carried(al2._h) > 0
    assert T.result_digest([x for g in got for x in g]) == T.result_digest(whole)


@pytest.mark.parametrize("carry", [0, 1, 12])
def test_gpu_host_batches_streamed(carry, gidx, monkeypatch):
    """bt_align_stream_submit / _collect: host batches handed over one after the other (three staging areas in HBM,
    PCIe on a copy stream), collected in order -- with carry-over a batch is collectable once its successor is in --
    each equal to the oracle's results."""
    import ctypes as C
    monkeypatch.setenv("BT_MAX_BLOCKS", "2")
    kw = T.MODES["n2_k3"]
    cap = 8
    al = aligner(gidx, "multi", kw)
    L = AL.lib()
    assert L.bt_ctx_set_carry(al._h, carry) == 0
    names = ["syn100", "syn36", "syn150", "syn50lowq", "syn76", "syn100"]      # syn150 (> 112 bases) cannot be carried: forces a flush
    jobs = []
    # every batch's reads in an order of this parameter's own: the three parameters run in one process, and a staging area
    # recycled from the context before must not hold what would be the right answer for this one (round 3's verdict)
    rng = np.random.default_rng(1000 + carry)
    for r in names:
        b0 = T.read_set("multi", r)
        perm = rng.permutation(b0.n)
        b = type(b0)(b0.seq[perm].copy(), b0.qual[perm].copy(), b0.len[perm].copy(), b0.seed[perm].copy(), [b0.names[i] for i in perm])
        k, rb = AL.pack_batch(b)
        hits = np.zeros(b.n * cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(b.n, dtype=np.uint32)
        status = np.zeros(b.n, dtype=np.uint8)
        pool = np.zeros(b.n * cap * 8, dtype=np.uint16)
        hb = A.HitBatchC(cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
        jobs.append(dict(b=b, keep=k, rb=rb, hits=hits, n_hits=n_hits, status=status, pool=pool, hb=hb))
    done = []
    tag = C.c_void_p()
    for i, j in enumerate(jobs):
        assert L.bt_align_stream_submit(al._h, C.byref(j["rb"]), C.byref(j["hb"]), C.c_void_p(i + 1)) == 0
        while True:                                          # whatever is complete by now, oldest first
            assert L.bt_align_stream_collect(al._h, C.byref(tag), 0) == 0
            if tag.value is None:
                break
            done.append(tag.value)
    while True:                                              # end of input: finish what is parked
        assert L.bt_align_stream_collect(al._h, C.byref(tag), 1) == 0
        if tag.value is None:
            break
        done.append(tag.value)
    assert done == list(range(1, len(jobs) + 1))
    pol = al.policy
    for r, j in zip(names, jobs):
        got = AL.unpack_hits(j["b"].n, cap, j["hits"], j["n_hits"], j["status"], j["pool"], int(pol.khits), int(pol.mhits), bool(pol.all_hits))
        want = T.oracle_results("multi", j["b"], kw, cap=cap)
        try:
            T.compare_results(got, want, "streamed %s carry=%d" % (r, carry))
        except AssertionError as e:
            # a rare failure of this test was seen once under six-process load: say as much as can be said
            bad = [i for i in range(j["b"].n) if got[i] != want[i]]
            raw = j["hits"].reshape(j["b"].n, cap)
            info = ["read %d n_hits %d status %d mm_off/nmm %s pool %s | got %r want %r" %
                    (i, int(j["n_hits"][i]), int(j["status"][i]), [(int(h["mm_off"]), int(h["nmm"])) for h in raw[i][:3]],
                     [hex(int(x)) for x in j["pool"][int(raw[i][0]["mm_off"]):int(raw[i][0]["mm_off"]) + 3]], got[i], want[i]) for i in bad[:6]]
            offs = sorted((int(h["mm_off"]), int(h["nmm"]), i) for i in range(j["b"].n) for h in raw[i][:min(cap, int(j["n_hits"][i]))] if int(h["nmm"]))
            overlaps = [(a, b) for a, b in zip(offs, offs[1:]) if a[0] + a[1] > b[0]][:6]
            raise AssertionError("%s\n%d reads differ: %s ...\nmismatch-list regions that overlap in the pool: %s\nmm_pool_used %d\n%s" %
                                 (e, len(bad), bad[:40], overlaps, int(j["hb"].mm_pool_used), "\n".join(info)))


def test_gpu_stream_room_counts_staging_areas(gidx, monkeypatch):
    """bt_align_stream_room: how many batches of a shape may ride as far as the device's memory goes -- half of what is free, in
    staging areas laid out as bt_align_stream_submit lays them out.  BT_FAKE_FREE_MB pretends the device has 100 MB free."""
    import ctypes as C
    monkeypatch.setenv("BT_FAKE_FREE_MB", "100")
    al = aligner(gidx, "e_coli", T.MODES["n2_k3"])
    L = AL.lib()
    n, stride, cap = 10000, 112, 8
    pool = n * cap * 6 + 1024
    rb = A.ReadBatchC(n, stride, 0, 0, 0, 0)
    hb = A.HitBatchC(cap, 0, 0, 0, 0, pool, 0)
    room = C.c_uint32(0)
    assert L.bt_align_stream_room(al._h, C.byref(rb), C.byref(hb), C.byref(room)) == 0
    r256 = lambda x: (x + 255) & ~255
    area = 2 * r256(n * stride) + r256(2 * n) + r256(4 * n) + r256(n * cap * 24) + r256(4 * n) + r256(n) + r256(2 * pool) + 256
    assert room.value == (50 << 20) // area and 5 < room.value < 20, (room.value, area)
    assert L.bt_align_stream_room(al._h, C.byref(rb), C.byref(hb), None) != 0          # BT_ERR_ARG
    hb0 = A.HitBatchC(0, 0, 0, 0, 0, pool, 0)
    assert L.bt_align_stream_room(al._h, C.byref(rb), C.byref(hb0), C.byref(room)) != 0


def test_gpu_stream_forty_batches_in_flight(gidx, monkeypatch):
    """bt_align_stream_submit with more batches in flight than rounds 2-5 had ring slots for (16): 40 host batches of 500
    reads handed over without waiting for any (512 lanes, reads ride along for up to 38 launches), collected as they
    complete and flushed at the end -- in order, each read's result the one the whole set gives as one batch."""
    import ctypes as C
    monkeypatch.setenv("BT_MAX_BLOCKS", "2")
    kw = T.MODES["n2"]
    cap = 1
    text = T.joined_text("e_coli")
    big = synth_reads(text, 20000, 76, mm_dist=(0, 1, 2, 2, 3, 4), seed=777)
    whole = aligner(gidx, "e_coli", kw).align(big, hit_cap=cap)
    al = aligner(gidx, "e_coli", kw)
    L = AL.lib()
    assert L.bt_ctx_set_carry(al._h, 38) == 0
    from bowtie_amd.reads import ReadBatch
    jobs = []
    for i in range(0, 20000, 500):
        b = ReadBatch(big.seq[i:i + 500].copy(), big.qual[i:i + 500].copy(), big.len[i:i + 500].copy(), big.seed[i:i + 500].copy(), big.names[i:i + 500])
        k, rb = AL.pack_batch(b)
        hits = np.zeros(b.n * cap, dtype=A.HIT_DTYPE)
        n_hits = np.zeros(b.n, dtype=np.uint32)
        status = np.zeros(b.n, dtype=np.uint8)
        pool = np.zeros(b.n * cap * 8, dtype=np.uint16)
        hb = A.HitBatchC(cap, hits.ctypes.data, n_hits.ctypes.data, status.ctypes.data, pool.ctypes.data, len(pool), 0)
        jobs.append(dict(b=b, keep=k, rb=rb, hits=hits, n_hits=n_hits, status=status, pool=pool, hb=hb))
    done, tag, most = [], C.c_void_p(), 0
    for i, j in enumerate(jobs):
        assert L.bt_align_stream_submit(al._h, C.byref(j["rb"]), C.byref(j["hb"]), C.c_void_p(i + 1)) == 0
        most = max(most, i + 1 - len(done))
        if i >= 24 and i % 8 == 7:                           # late, and now and then: whatever is complete by now, oldest first
            while True:
                assert L.bt_align_stream_collect(al._h, C.byref(tag), 0) == 0
                if tag.value is None:
                    break
                done.append(tag.value)
    while True:
        assert L.bt_align_stream_collect(al._h, C.byref(tag), 1) == 0
        if tag.value is None:
            break
        done.append(tag.value)
    assert done == list(range(1, len(jobs) + 1))
    assert most > 16, most                                   # the point of the test
    pol = al.policy
    got = []
    for j in jobs:
        got += AL.unpack_hits(j["b"].n, cap, j["hits"], j["n_hits"], j["status"], j["pool"], int(pol.khits), int(pol.mhits), bool(pol.all_hits))
    assert T.result_digest(got) == T.result_digest(whole)


# ---- the best-first engine (--best, --strata, -M, -v 3): bt_best_kernel ---------------------------
BEST_RAGGED = ["n2_best", "v3", "v2_a_best_strata", "n3_best", "n2_M3", "v1_best", "n1_best", "n0_best_a_m3",
               "n3_best_a_l12_e200", "n2_k2_best_strata_m5"]


@pytest.mark.parametrize("mode", BEST_RAGGED)
def test_gpu_best_first_vs_oracle_ragged(mode, gidx):
    """Ragged 1..150-base reads with Ns and low qualities: hits and op counts equal the oracle's."""
    kw = T.MODES[mode]
    text = T.joined_text("multi")
    rng = np.random.default_rng(11)
    reads = []
    for i in range(1500):
        L = int(rng.integers(1, 151))
        b = synth_reads(text, 1, L, mm_dist=(0, 1, 2, 3), seed=11000 + i, n_frac=0.2, lowq_frac=0.1)
        reads.append(Read(("q%d" % i).encode(), b.seq[0, :L].copy(), b.qual[0, :L].tobytes()))
    batch = pack_reads(reads)
    import oracle_lib as OL
    oc, gc = OL.OpCounts(), A.OpCounts()
    want = T.oracle_results("multi", batch, kw, cap=T.hit_cap_for(kw), counts=oc)
    got = aligner(gidx, "multi", kw).align(batch, hit_cap=T.hit_cap_for(kw), counts=gc)
    T.compare_results(got, want, mode)
    T.check_op_counts(oc, gc)


@pytest.mark.parametrize("mode,length,n", [("n2_best", 100, 20000), ("v3", 76, 20000), ("v2_a_best_strata", 50, 10000)])
def test_gpu_best_first_vs_oracle_e_coli_synthetic(mode, length, n, gidx):
    kw = T.MODES[mode]
    batch = synth_reads(T.joined_text("e_coli"), n, length, seed=777 + length)
    want = T.oracle_results("e_coli", batch, kw, cap=T.hit_cap_for(kw))
    got = aligner(gidx, "e_coli", kw).align(batch, hit_cap=T.hit_cap_for(kw))
    T.compare_results(got, want, mode)


def test_gpu_best_first_arena_overflow_retry(gidx, monkeypatch):
    """Reads that outgrow a (deliberately tiny) arena are re-run through the twin context."""
    monkeypatch.setenv("BT_BEST_ARENA_WORDS", "1200")
    kw = T.MODES["n2_best"]
    batch = T.read_set("multi", "syn100")
    al = aligner(gidx, "multi", kw)
    got = al.align(batch)
    assert al.last_retried > 0
    T.compare_results(got, T.oracle_results("multi", batch, kw), "n2_best tiny arenas")


def test_gpu_best_first_with_fewer_arenas_than_lanes(gidx, monkeypatch, capfd):
    """A best-first context takes at most 45 % of the device's free memory for its arenas and runs with fewer lanes when that is
    not enough for one arena per lane (several contexts on one GPU: bowtie-amd --inflight, six test processes).  BT_FAKE_FREE_MB
    pretends the device has 600 MB free: 45 % of that are arenas for 1 024 of the 4 096 lanes this batch could use; the
    results are the oracle's all the same, and BT_VERBOSE says what was allocated."""
    monkeypatch.setenv("BT_FAKE_FREE_MB", "600")
    monkeypatch.setenv("BT_VERBOSE", "1")
    batch = T.read_set("multi", "syn100")
    for mode in ("n2_best", "v2_a_best_strata"):
        kw = T.MODES[mode]
        reads = type(batch)(np.tile(batch.seq, (7, 1)), np.tile(batch.qual, (7, 1)), np.tile(batch.len, 7), np.tile(batch.seed, 7), batch.names * 7)
        al = aligner(gidx, "multi", kw)
        got = al.align(reads, hit_cap=T.hit_cap_for(kw))
        want = T.oracle_results("multi", batch, kw, cap=T.hit_cap_for(kw))
        T.compare_results(got, want * 7, "fewer arenas " + mode)
        err = capfd.readouterr().err
        assert "arenas for 1024 lanes (4352 asked)" in err or "arenas for 1024 lanes" in err, err[-600:]


def test_gpu_best_first_large_batch_properties(gidx):
    """300 k reads x 100 bp -n 2 --best: idempotence, permutation equivariance, and a sample
    against the oracle."""
    text = T.joined_text("e_coli")
    n = 300_000
    batch = synth_reads(text, n, 100, mm_dist=(0, 1, 2, 2, 3, 4), seed=4321)
    kw = T.MODES["n2_best"]
    al = aligner(gidx, "e_coli", kw)
    r1 = al.align(batch)
    assert T.result_digest(r1) == T.result_digest(al.align(batch))
    from bowtie_amd.reads import ReadBatch
    idx = np.arange(0, n, 149)
    sub = ReadBatch(batch.seq[idx], batch.qual[idx], batch.len[idx], batch.seed[idx], [batch.names[i] for i in idx])
    want = T.oracle_results("e_coli", sub, kw)
    T.compare_results([r1[i] for i in idx], want, "sample of large batch")
    perm = np.random.default_rng(3).permutation(len(idx))
    pb = ReadBatch(sub.seq[perm], sub.qual[perm], sub.len[perm], sub.seed[perm], [sub.names[i] for i in perm])
    rp = al.align(pb)
    T.compare_results(rp, [want[i] for i in perm], "permuted")


# ---- paired-end (bt_align_pairs: PairedBWAlignerV2 + reference window scan) -------------------------
@pytest.mark.parametrize("run", T.paired_runs(), ids=lambda r: r["file"][:-7])
def test_gpu_paired_matches_reference_sam(run, gidx):
    b1, b2 = T.pair_set(run["index"], run["reads"])
    kw = T.MODES[run["mode"]]
    al = aligner(gidx, run["index"], kw)
    res = al.align_pairs(b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    T.check_pairs_against_golden(run, res, b1, b2, gidx[run["index"]].refnames)


@pytest.mark.parametrize("run", T.paired_v1_runs(), ids=lambda r: r["file"][6:-7])
def test_gpu_paired_without_best_matches_reference_sam(run, gidx):
    """Paired-end without --best (PairedBWAlignerV1) on the GPU against the reference's outputs."""
    b1, b2 = T.pair_set(run["index"], run["reads"])
    kw = dict(T.MODES[run["mode"]], pe_v1=True)
    al = aligner(gidx, run["index"], kw)
    res = al.align_pairs(b1, b2, hit_cap=2048 if kw.get("all_hits") else None)
    T.check_pairs_against_golden(run, res, b1, b2, gidx[run["index"]].refnames)


@pytest.mark.parametrize("mode", ["pe_n1_best_X500", "pe_n2_best_X400_I250_k3", "pe_v3_best_X500", "pe_n1_a_strata_X500"])
def test_gpu_paired_vs_oracle_counts(mode, gidx):
    import oracle_lib as OL
    kw = T.MODES[mode]
    b1, b2 = T.pair_set("multi", "pe50")
    oc, gc = OL.OpCounts(), A.OpCounts()
    cap = 2048 if kw.get("all_hits") else None
    want = T.oracle_pair_results("multi", b1, b2, kw, cap=cap, counts=oc)
    got = aligner(gidx, "multi", kw).align_pairs(b1, b2, hit_cap=cap, counts=gc)
    T.compare_results(got, want, mode)
    T.check_op_counts(oc, gc)


def test_gpu_paired_config5_shape(gidx):
    """BASELINE config 5's shape on e_coli: 2 x 50 bp pairs, -n 1 --best -X 500: a 20 k-pair sample
    against the oracle, idempotence."""
    from bowtie_amd.synth import synth_pairs
    kw = T.MODES["pe_n1_best_X500"]
    b1, b2 = synth_pairs(T.joined_text("e_coli"), 20000, 50, seed=5050)
    al = aligner(gidx, "e_coli", kw)
    r1 = al.align_pairs(b1, b2)
    assert T.result_digest(r1) == T.result_digest(al.align_pairs(b1, b2))
    from bowtie_amd.reads import ReadBatch
    idx = np.arange(0, 20000, 13)
    s1 = ReadBatch(b1.seq[idx], b1.qual[idx], b1.len[idx], b1.seed[idx], [b1.names[i] for i in idx])
    s2 = ReadBatch(b2.seq[idx], b2.qual[idx], b2.len[idx], b2.seed[idx], [b2.names[i] for i in idx])
    T.compare_results([r1[i] for i in idx], T.oracle_pair_results("e_coli", s1, s2, kw), "config-5 shape")
    assert sum(1 for h, _, _ in r1 if h) > 15000


def test_gpu_bench_two_ranks_equal_one(tmp_path):
    """bench.py's N > 1 path (init_process_group, read sharding, the int64 counter all-reduce): two ranks
    sharing the one GPU over gloo, strong scaling, must report the same hit counters as one rank over
    the same read pool."""
    import json
    import subprocess
    import sys
    bench = os.path.join(T.ROOT, "bench.py")
    common = ["--workload", "ecoli_n2_100", "--reads", "300000", "--steps", "1", "--warmup", "0", "--no-cpu",
              "--scaling", "strong", "--no-verify"]
    one = subprocess.run([sys.executable, bench, "--gpus", "1"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", bench, "--gpus", "2",
                          "--dist-backend", "gloo"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    j1 = json.loads(one.stdout.decode().strip().split("\n")[-1])
    j2 = json.loads(two.stdout.decode().strip().split("\n")[-1])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert j1["config"]["hit_counters_last_step"] == j2["config"]["hit_counters_last_step"]
    assert j1["config"]["hit_counters_last_step"]["aligned"] > 200000


# ---- the wavefront automaton (bt_best_kernel) on the small indexes ------------------------------------------------------
# The library picks the best-first loop by index size: on e_coli / multi the tests above run the call-by-call kernel
# (bt_best_nested_kernel).  BT_BEST_NESTED=0 (read at every launch) puts the same inputs through the automaton.
from best_modes import BEST_MODES


@pytest.fixture
def automaton(monkeypatch):
    monkeypatch.setenv("BT_BEST_NESTED", "0")


@pytest.mark.parametrize("run", [r for r in T.golden_runs() if r["mode"] in BEST_MODES], ids=lambda r: r["file"][:-7])
def test_gpu_automaton_matches_reference_sam(run, gidx, automaton):
    test_gpu_matches_reference_sam(run, gidx)


@pytest.mark.parametrize("run", T.paired_runs(), ids=lambda r: r["file"][:-7])
def test_gpu_automaton_paired_matches_reference_sam(run, gidx, automaton):
    test_gpu_paired_matches_reference_sam(run, gidx)


@pytest.mark.parametrize("mode", BEST_RAGGED)
def test_gpu_automaton_best_first_vs_oracle_ragged(mode, gidx, automaton):
    test_gpu_best_first_vs_oracle_ragged(mode, gidx)


@pytest.mark.parametrize("mode", ["pe_n1_best_X500", "pe_n2_best_X400_I250_k3", "pe_v3_best_X500", "pe_n1_a_strata_X500"])
def test_gpu_automaton_paired_vs_oracle_counts(mode, gidx, automaton):
    test_gpu_paired_vs_oracle_counts(mode, gidx)


def test_gpu_automaton_large_batches(gidx, automaton):
    test_gpu_best_first_large_batch_properties(gidx)
    test_gpu_paired_config5_shape(gidx)


def test_gpu_automaton_arena_overflow_retry(gidx, monkeypatch):
    monkeypatch.setenv("BT_BEST_NESTED", "0")
    test_gpu_best_first_arena_overflow_retry(gidx, monkeypatch)
