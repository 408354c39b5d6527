"""How a job of chunks is cut into passes of the hot path (host logic only: no GPU, no torch).

The reference decodes one VAD segment after the other (whisperx/backends/mlx_lightning.py:82-119) or `batch_size`
segments per call (whisperx/asr.py:80-87); here the backend's scheduler decides how many rows a pass carries and how
many passes are in flight, because on MI355X that is what sets the bytes a chunk costs (the decoder weights are
streamed once per pass) and how the passes share the HBM.  Rows are independent and every reduction in the kernels has a
fixed order, so the cut changes no token (tests/test_gpu_backend.py::test_a_chunk_decodes_the_same_in_every_job).
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

DEFAULT_ROWS = 128   # rows of the contexts the default scheduler works with (plan_passes)
MAX_ROWS = 128       # rows an engine context takes at most (wx_create; ~38 GB of workspace per context at 128 rows of large-v3)


def pass_sizes(n_chunks: int, rows_per_pass: int, lanes: int) -> List[int]:
    """Rows of each pass for `n_chunks` chunks in passes of <= `rows_per_pass` (R), pass i running on context i % lanes.

    Whole rounds of full passes first -- `lanes` passes of R rows each, as many rounds as fit -- then the remaining
    M < lanes * R chunks as ONE more round of equal passes: as many as there are contexts, unless that would make them
    smaller than 8 rows.  A pass costs its decoder weights and its launch chain whatever its rows (about 5 rows' worth),
    so tiny passes are all overhead, while a remainder cut into full passes leaves contexts idle: 100 chunks on 4
    contexts run as 4 x 16 then 4 x 9 (not 6 x 16 + 4), 81 chunks as 4 x 16 then 9 + 8, 5 chunks as one pass.  Every
    context then carries the same number of passes (one fewer for some in the last round) and about the same rows --
    the least makespan a per-pass cost of a + b * rows allows -- and the full-R launch shape, whose hipGraphs every job
    of >= lanes * R chunks captures first, serves every round but the last."""
    lanes = max(1, lanes)
    R = max(1, rows_per_pass)
    if n_chunks <= 0:
        return [0]
    rounds, rest = divmod(n_chunks, lanes * R)
    sizes = [R] * (rounds * lanes)
    if rest:
        least = max(1, min(8, R // 2))                              # rows of the smallest pass worth its fixed cost
        n_tail = max(-(-rest // R), min(lanes, rest // least))
        sizes += [rest // n_tail + (1 if i < rest % n_tail else 0) for i in range(n_tail)]
    return sizes


def plan_passes(n_chunks: int, rows_cap: int, lanes_16: int = 4, lanes_wide: int = 3, launch_bound: bool = False) -> Tuple[List[int], int]:
    """(rows of each pass in launch order, passes in flight) for a job of `n_chunks` chunks on contexts that take up to
    `rows_cap` rows.

    What a pass costs (large-v3, tools/ab_rows_inflight.py, ms per 16 chunks in steady state): 16 rows x 4 in flight 204,
    32 x 3 189, 48 x 3 186, 64 x 3 181.5, 64 x 2 184.5, 64 x 1 226, 128 x 3 ~165 -- a pass streams the decoder weights
    once whatever its rows (49 MB of 172 MB per layer at 16 rows), and wider cross-attention launches stream better
    (6.2 TB/s at 128 rows against 4.8 at 16), but one pass alone leaves the HBM idle during its GEMV chain.  A launch
    costs its GEMV chain per GROUP of 16 rows, so rows that do not fill their group are paid for in full (tools/ab_plan.py:
    320 chunks as 5 x 64 2 629x, as 6 x 53-54 2 499x).  And every pass of a job decodes the same number of steps, so a
    narrower pass ends earlier and leaves the others two in flight (320 chunks as 64 + 128 + 128: the 64-row pass lands
    800 ms before the others).  So: the job's groups of 16 rows are dealt evenly to `lanes_wide` contexts, each context's
    share is cut into passes of <= rows_cap rows, as equal as whole groups allow, and the passes are issued round by
    round (pass i runs on context i % lanes); the ragged group comes off the first pass.  320 chunks: 112 + 112 + 96
    (2 861x against 2 834x for 64 + 128 + 128); 400: 80 + 128 + 128 + 64 (context 0: 80 then 64); 100: 36 + 32 + 32;
    81: 17 + 32 + 32.  Jobs too small for three passes of more than 16 rows, and jobs that fit ONE round of 16-row passes on
    the `lanes_16` contexts (<= 64 chunks on four: a 30-minute file as 4 x 15), are cut by pass_sizes() into <= 16-row
    passes on up to `lanes_16` contexts.

    launch_bound (models of d <= 512: whisper-tiny / base, BASELINE.json config 2): a decode step of such a model is a chain
    of ~40 dependent launches over a few MB per row -- its length is the launch chain, not the bytes, and passes in flight
    on several hardware queues stretch every launch's dispatch (5-8 us against ~2 us alone).  A small job then runs best as
    ONE pass: whisper-tiny, 60 chunks: 1 x 60 60.1 ms (29 970x), 2 x 30 67.7, 3 x 20 75.1, 4 x 15 78.6 (the rule above),
    profiles/r05_tiny_plans.txt; from ~100 chunks on the default cut is as good as any (120 chunks: 105 ms against 110)."""
    if launch_bound and n_chunks <= min(rows_cap, 64):
        return [n_chunks], 1
    # small jobs: <= 16-row passes on up to `lanes_16` contexts.  Up to one full round of them (4 x 16 = 64 chunks) that beats
    # three wider passes: a 30-minute file, 60 chunks, runs 2 358x as 4 x 15 against 2 304x as 28 + 16 + 16 (tools/ab_small_jobs.py)
    if rows_cap <= 16 or n_chunks < 3 * 16 + 1 or n_chunks <= 16 * min(lanes_16, 4):
        R = max(1, min(rows_cap, 16))
        lanes = max(1, min(lanes_16, -(-n_chunks // R)))
        return pass_sizes(n_chunks, R, lanes), lanes
    lanes = max(1, lanes_wide)
    cap_units = max(1, rows_cap // 16)
    units = -(-n_chunks // 16)
    per_lane = [units // lanes + (1 if i < units % lanes else 0) for i in range(lanes)]
    lane_passes = []
    for u in per_lane:
        k = -(-u // cap_units) if u else 0
        lane_passes.append(sorted((u // k + (1 if i < u % k else 0) for i in range(k)), reverse=True) if k else [])
    sizes = [lane_passes[l][d] * 16 for d in range(max(len(p) for p in lane_passes)) for l in range(lanes) if d < len(lane_passes[l])]
    if n_chunks % 16:
        sizes[0] -= 16 - n_chunks % 16
    return [r for r in sizes if r > 0], lanes


def launch_shape(sizes: List[int], cap: int) -> int:
    """R, the row count the job's launches are captured for.  Passes of <= 16 rows are all launched with min(cap, 16)
    rows (padding rows count as finished, wx_decode_opts.n_active: one hipGraph set whatever the remainder); wider passes
    launch whole groups of 16 rows (the GEMV kernels walk those), so R is the widest pass rounded up to its groups."""
    widest = max(sizes) if sizes else 1
    return min(cap, 16) if widest <= 16 else min(cap, 16 * -(-widest // 16))


@dataclass
class JobPlan:
    sizes: List[int]          # rows of each pass, in launch order (pass i runs on context i % lanes)
    R: int                    # launch shape of the job (launch_shape)
    lanes: int                # passes in flight

    def launch_rows(self, n: int) -> int:
        """rows a pass of n chunks is launched with: R for passes of <= 16 rows, its whole 16-row groups otherwise"""
        r = self.R if self.R <= 16 else min(self.R, 16 * -(-n // 16))
        assert r >= n, (n, r, self.R, self.sizes)
        return r

    def report(self):
        n_pass = len(self.sizes)
        return {"rows": list(self.sizes), "launch_rows": self.launch_rows(max(self.sizes)) if self.sizes else self.R,
                "passes_in_flight": max(1, min(self.lanes, n_pass))}


def plan_job(n_chunks: int, cap: int, lanes_for: Callable[[int, int], int], auto_rows: bool = True,
             rows_per_pass: Optional[int] = None, passes_in_flight: Optional[int] = None,
             pass_rows: Optional[List[int]] = None, default_rows: Optional[int] = None, launch_bound: bool = False) -> JobPlan:
    """The cut of one scheduler run.

    cap: rows the engine contexts take; lanes_for(R, need) -> passes that can really be in flight at launch shape R when
    the job has `need` passes (the backend asks its streams, WhisperHipBackend._default_lanes); auto_rows: the default
    scheduler (plan_passes) unless the caller pins rows_per_pass / passes_in_flight / an explicit `pass_rows` cut.

    When fewer lanes are available than the plan wanted (streams that share a hardware queue, a context that did not
    fit in memory) the job is cut again for those lanes -- and R is taken from THAT cut: two lanes make wider passes
    than three (100 chunks: 36 + 32 + 32 becomes 52 + 48), so the launch shape grows with them (R used to
    stay at the first plan's 48, the launcher then captured a 52-row shape the pre-warm did not know about)."""
    cap = max(1, cap)
    R0 = max(1, min(rows_per_pass or default_rows or cap, cap))
    if pass_rows:                            # an explicit cut (tools/ab_plan.py): rows of every pass, dealt round-robin to the contexts
        assert sum(pass_rows) == n_chunks and max(pass_rows) <= cap, (pass_rows, n_chunks, cap)
        sizes = list(pass_rows)
        R = min(cap, 16 * -(-max(sizes) // 16))
        lanes = passes_in_flight or lanes_for(R, len(sizes))
        return JobPlan(sizes, R, max(1, lanes))
    if auto_rows and not rows_per_pass and not passes_in_flight:
        sizes, want = plan_passes(n_chunks, R0, launch_bound=launch_bound)
        R = launch_shape(sizes, R0)
        lanes = max(1, lanes_for(R, want))
        if lanes < want:                     # fewer streams run side by side than the plan assumed: cut for those
            sizes, _ = plan_passes(n_chunks, cap, lanes_16=lanes, lanes_wide=lanes, launch_bound=launch_bound)
            R = launch_shape(sizes, cap)
        return JobPlan(sizes, R, lanes)
    lanes = passes_in_flight or lanes_for(R0, max(1, -(-n_chunks // R0)))
    return JobPlan(pass_sizes(n_chunks, R0, max(1, lanes)), R0, max(1, lanes))
