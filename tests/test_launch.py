"""`--gpus N` of bench.py / tools/bench_recurrent.py starts its own ranks (puresound_amd/launch.py): CPU tests -- the ranks
leave through PS_LAUNCH_PROBE before any GPU call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{") and '"probe"' in l]


@pytest.mark.parametrize("script", ["bench.py", os.path.join("tools", "bench_recurrent.py")])
def test_gpus_flag_starts_its_own_ranks(script):
    """`python bench.py --gpus 2` with no outer launcher: the parent (which never touches a GPU) starts two ranks through
    torch.distributed.run on 127.0.0.1; PS_LAUNCH_PROBE makes each rank report its environment and leave before any GPU
    call, so this runs on the CPU box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PS_LAUNCH_PROBE"] = "1"
    extra = ["--which", "cfg4"] if "recurrent" in script else []
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2", "--steps", "1", "--warmup", "0", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    probes = _probe_lines(out.stdout)
    assert sorted(p["rank"] for p in probes) == [0, 1]
    assert all(p["world_size"] == 2 and p["master_addr"] == "127.0.0.1" for p in probes)


def test_gpus_flag_must_match_an_outer_launch():
    """Under an outer launcher the flag still has to agree with the world it was given."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_a_failing_rank_fails_the_self_launched_run():
    sys.path.insert(0, ROOT)
    from puresound_amd import launch
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    try:
        assert launch.self_launch(f.name, [], 2) != 0
    finally:
        os.unlink(f.name)


def test_design_md_round_numbers_are_the_generated_ones():
    """DESIGN.md section 5 carries the round's headline numbers between markers; they are generated from the committed profile
    files by tools/design_numbers.py and may not be edited by hand ('DESIGN numbers from CSVs')."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("design_numbers", os.path.join(root, "tools", "design_numbers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(root, "DESIGN.md")).read()
    a = text.index("<!-- numbers:r04")
    a = text.index("\n", a) + 1
    b = text.index("<!-- /numbers:r04 -->")
    assert text[a:b].rstrip("\n") == mod.block("r04")
