"""CPU: whisperx_mlx_amd/scheduler.py -- how a job is cut into passes, and the launch shape that goes with the cut.

The branch ADVICE r03 found: when fewer lanes are available than plan_passes wanted (streams that collide on the default
4 hardware queues, a 2nd / 3rd context that did not fit in memory) the job is cut again for those lanes; the launch
shape R must come from THAT cut (two lanes make wider passes than three)."""
import pytest

from whisperx_mlx_amd import scheduler as S


def _lanes(avail):
    calls = []

    def f(R, need):
        calls.append((R, need))
        return max(1, min(avail, need))
    f.calls = calls
    return f


@pytest.mark.parametrize("avail", [1, 2, 3])
def test_replanned_jobs_launch_every_pass_with_enough_rows(avail):
    """every (n, lanes) combination the advisor simulated (n = 49..599, lanes 1 and 2 gave 711 passes wider than the
    stale R): after the re-plan every pass fits its launch, R is a whole number of 16-row groups, and R <= cap"""
    bad = []
    for n in list(range(1, 700)) + [768, 1221, 4800]:
        plan = S.plan_job(n, 128, _lanes(avail))
        assert sum(plan.sizes) == n and all(0 < r <= 128 for r in plan.sizes), (n, plan)
        assert plan.lanes <= avail or plan.lanes == 1
        for r in plan.sizes:
            lr = plan.launch_rows(r)            # asserts lr >= r itself
            if lr < r or lr > 128 or (plan.R > 16 and lr % 16):
                bad.append((n, r, lr, plan.R))
        rep = plan.report()
        assert rep["launch_rows"] >= max(plan.sizes) and rep["rows"] == plan.sizes
        assert rep["passes_in_flight"] == max(1, min(plan.lanes, len(plan.sizes)))
    assert not bad, bad[:5]


def test_the_advisors_example():
    """100 chunks plan as 36 + 32 + 32 with R = 48; on two lanes they become 52 + 48 and must launch with 64 rows"""
    want3 = S.plan_job(100, 128, _lanes(3))
    assert want3.sizes == [36, 32, 32] and want3.R == 48 and want3.lanes == 3
    two = S.plan_job(100, 128, _lanes(2))
    assert two.sizes == [52, 48] and two.lanes == 2
    assert two.R == 64 and two.launch_rows(52) == 64 and two.launch_rows(48) == 48
    one = S.plan_job(100, 128, _lanes(1))
    assert one.sizes == [100] and one.R == 112 and one.launch_rows(100) == 112


def test_plan_job_other_modes():
    # pinned rows per pass / passes in flight: pass_sizes, launched with the pinned rows
    p = S.plan_job(81, 128, _lanes(4), rows_per_pass=16, passes_in_flight=4)
    assert p.sizes == S.pass_sizes(81, 16, 4) and p.R == 16 and p.launch_rows(9) == 16
    p = S.plan_job(100, 128, _lanes(3), rows_per_pass=48)
    assert p.R == 48 and p.launch_rows(33) == 48 and p.launch_rows(16) == 16 and sum(p.sizes) == 100
    # coalesce=k backends (auto_rows False): the context's rows are the default
    p = S.plan_job(100, 64, _lanes(3), auto_rows=False, default_rows=48)
    assert p.R == 48 and max(p.sizes) <= 48
    # an explicit cut
    p = S.plan_job(320, 128, _lanes(3), pass_rows=[64, 128, 128])
    assert p.sizes == [64, 128, 128] and p.R == 128 and p.launch_rows(64) == 64
    with pytest.raises(AssertionError):
        S.plan_job(320, 128, _lanes(3), pass_rows=[64, 128, 129])
    # small jobs: <= 16-row passes on up to four contexts, one launch shape
    f = _lanes(4)
    p = S.plan_job(40, 128, f)
    assert p.R == 16 and p.lanes == 3 and p.sizes == [14, 13, 13] and f.calls == [(16, 3)]
    assert S.plan_job(5, 128, _lanes(4)).sizes == [5]
    # contexts smaller than 16 rows
    p = S.plan_job(10, 4, _lanes(4))
    assert p.R == 4 and max(p.sizes) <= 4 and p.launch_rows(3) == 4
    # the driver's bench job and the launch shapes it needs
    p = S.plan_job(320, 128, _lanes(3))
    assert p.sizes == [112, 112, 96] and p.R == 112 and [p.launch_rows(r) for r in p.sizes] == [112, 112, 96]


def test_launch_shape():
    assert S.launch_shape([5], 128) == 16 and S.launch_shape([16, 9], 128) == 16 and S.launch_shape([17], 128) == 32
    assert S.launch_shape([117, 96], 128) == 128 and S.launch_shape([3], 2) == 2 and S.launch_shape([40], 40) == 40


def test_launch_bound_models_run_small_jobs_as_one_pass():
    """whisper-tiny / base (d <= 512): a decode step is a launch chain, and passes in flight stretch every dispatch -- a job of
    <= 64 chunks is ONE pass (config 2's 60 chunks: 60.1 ms against 78.6 as 4 x 15, profiles/r05_tiny_plans.txt); larger
    jobs, and every job of the wide models, are cut as before"""
    from whisperx_mlx_amd.scheduler import plan_job, plan_passes
    assert plan_passes(60, 128, launch_bound=True) == ([60], 1)
    assert plan_passes(30, 128, launch_bound=True) == ([30], 1)
    assert plan_passes(64, 128, launch_bound=True) == ([64], 1)
    assert plan_passes(60, 48, launch_bound=True) == plan_passes(60, 48)          # the context does not take 60 rows: the usual cut
    assert plan_passes(65, 128, launch_bound=True) == plan_passes(65, 128)
    assert plan_passes(320, 128, launch_bound=True) == ([112, 112, 96], 3)
    assert plan_passes(60, 128) == ([15, 15, 15, 15], 4)
    p = plan_job(60, 128, lambda R, need: 3, launch_bound=True)
    assert p.sizes == [60] and p.R == 64 and p.lanes >= 1 and p.launch_rows(60) == 64
