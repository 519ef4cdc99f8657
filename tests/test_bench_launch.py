"""`python bench.py --gpus N` as the driver calls it (no launcher in front) must start its own ranks.

On a box without a GPU every measuring rank stops at "needs an MI355X" (rc 3) -- what this checks is that the ranks WERE
started (rank supervisors through torch.distributed.run on 127.0.0.1, each starting its measuring child per transport
attempt) and that the failure is relayed -- on stderr and as ONE stdout line with `value` null --, instead of the rc 2 "launch me
under torchrun" refusal of round 1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("covered on the GPU by test_gpu_two_ranks.py::test_bench_script_multi_rank_path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode not in (0, 2), (p.returncode, p.stderr[-2000:])
    # both ranks ran main() -- once per transport of the default chain (rccl, rccl-plain -- the RCCL transport's plain form,
    # tried because rccl gave no number --, ipc, host): the rank supervisors start a fresh pair of children for every attempt
    # (a supervisor ends its child as soon as ANY rank's child has failed: between one and two messages per attempt)
    assert 4 <= p.stderr.count("bench.py needs an MI355X") <= 2 * 4, p.stderr[-3000:]
    assert "starting fresh ranks on rccl-plain" in p.stderr and "starting fresh ranks on ipc" in p.stderr
    assert "starting fresh ranks on host" in p.stderr and "no transport left" in p.stderr
    # no number -- and still exactly ONE line on stdout, the contract's fields with `value` null and what happened (gloo's
    # "[Gloo] Rank 0 is connected to ..." goes to stderr: the supervisors set their group up with fd 1 pointing there)
    import json

    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_gpus"] == 2 and rec["roofline"] is None and rec["cpu_baseline"] is None
    assert rec["error"].startswith("no transport of the chain produced a result: rccl: ")
    assert [f["transport"] for f in rec["transport_fallback"]] == ["rccl", "rccl-plain", "ipc", "host"]
    assert len(lines[0]) <= 8000
