"""CPU: the host-side work planner of the fused kernel (prf_vertical_plan through prf_plan_describe).
Every motif size must be scanned exactly once, by the right kind of task, and a chunk may skip aligned groups
(stride 2 / 4) only if every motif size in it needs runs long enough to contain an examined group."""
import itertools

import prf_native


def M(k, r, span):
    return max((r - 1) * k, span - k)


def check(kmin, kmax, r, span):
    plan = prf_native.plan_describe(kmin, kmax, r, span)
    if kmax > 480:
        assert plan["path"] == "generic"
        return plan
    assert plan["path"] == "fused" and 1 <= plan["waves"] <= 4
    seen = {}
    for t in plan["tasks"]:
        assert 0 <= t["wave"] < plan["waves"]
        if t["kind"] == 0:
            assert t["k0"] % 4 == 0 and t["stride"] in (1, 2, 4)
            ks = [t["k0"] + i for i in range(8) if (t["valid"] >> i) & 1]
            assert ks
            for k in ks:
                assert M(k, r, span) >= 15                       # groups of 8 rows need runs of >= 15
                assert M(k, r, span) >= 8 * t["stride"] + 7      # a run that long contains an examined group
        else:
            ks = [t["k0"]]
            assert t["kind"] == M(t["k0"], r, span) < 15          # exact task, templated on M
        for k in ks:
            assert k not in seen, (k, t, seen[k])
            seen[k] = t
    assert sorted(seen) == list(range(kmin, kmax + 1))
    # the LDS image must be wide enough for the furthest row of a lane's extended stream any task reads: a group task
    # reads rows 24 + k0 .. 24 + k0 + 15 in its last block, an exact task the whole 16-byte slots of rows 0 .. 31 + M - 1 + k
    reach = max([24 + t["k0"] + 15 for t in plan["tasks"] if t["kind"] == 0] +
                [4 * ((32 + t["kind"] - 1 + t["k0"] + 3) // 4) - 1 for t in plan["tasks"] if t["kind"]])
    assert plan["nc"] >= 64 + reach // 32 and plan["lds_bytes"] <= 160 * 1024 // 3
    for t in plan["tasks"]:
        if t["kind"]:
            assert t["k0"] <= t["kind"] <= 14        # exact tasks are compiled per (k, M), k <= M < 15
    return plan


def test_default_parameters():
    plan = check(1, 50, 3, 9)
    assert plan["waves"] == 4
    strides = {t["k0"]: t["stride"] for t in plan["tasks"] if t["kind"] == 0}
    # (round 3: a task may take four sizes only, so that 12-15 are examined at every 2nd group and 20-23 at every 4th)
    assert strides == {8: 1, 12: 2, 20: 4, 28: 4, 36: 4, 44: 4}
    assert {t["k0"]: t["valid"] for t in plan["tasks"] if t["kind"] == 0} == {8: 0x0F, 12: 0xFF, 20: 0xFF, 28: 0xFF, 36: 0xFF, 44: 0x7F}
    assert sorted(t["k0"] for t in plan["tasks"] if t["kind"]) == [1, 2, 3, 4, 5, 6, 7]


def test_parameter_sweep():
    for kmin, width, r, span in itertools.product((1, 2, 3, 5, 9, 16, 33), (0, 1, 5, 17, 50, 130), (2, 3, 4, 7), (1, 5, 9, 14, 15, 16, 30, 200)):
        check(kmin, kmin + width, r, span)
    check(1, 480, 3, 9)
    check(300, 481, 2, 5)


def test_whole_contig_shares_cover_the_genome_in_order():
    """multi_gpu.plan_whole_contigs (min_repeats == 1 under N ranks): every contig exactly once, whole, ranks in genome order
    (so the ranks' BED pieces concatenate), lengths balanced as far as whole contigs allow."""
    import multi_gpu
    hg38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
            135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
            50818468, 156040895, 57227415, 16569]
    for lens in (hg38, [5], [0, 0, 7, 0], [], [10] * 3):
        for world in (1, 2, 3, 8):
            shares = multi_gpu.plan_whole_contigs(lens, world)
            assert len(shares) == world
            flat = [p for share in shares for p in share]
            assert flat == [(c, 0, n) for c, n in enumerate(lens)]          # whole contigs, genome order across the ranks
    loads = [sum(e for _c, _b, e in share) for share in multi_gpu.plan_whole_contigs(hg38, 8)]
    assert max(loads) < 1.5 * sum(hg38) / 8
