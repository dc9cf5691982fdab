"""bench.py starts its own rank processes: `python bench.py --gpus 2` with no launcher.  Two ranks
share the one GPU of the test box, so the reduction backend is gloo and the C5 exchange goes through torch.distributed
(a rehearsal of the orchestration, not an xGMI measurement)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_self_launch_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--blocks", "2", "--prewarm-ms", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    c5 = out["c5_sharded"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["value"] > 0 and c5["parity_max_rel_vs_oracle"] < 1e-4
