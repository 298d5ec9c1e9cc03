"""bench.py launch plumbing on CPU (no GPU, no HIP library): `--gpus N` without a launcher starts N ranks itself,
rank 0 prints ONE JSON line labelled with the real world size, and a launcher/flag mismatch is an error."""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_gpus_2_self_launches_two_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line (rank 0)
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak" and j["dry_run"] is True
    assert j["unit"] == "Msamples/s" and j["higher_is_better"] is True and "roofline" in j
    assert j["config"]["channels_per_gpu"] == 65536
    # value = samples of ALL ranks / wall time
    assert abs(j["value"] - 2 * 65536 * 128 * 3 / (j["ms_per_step"] * 3e-3) / 1e6) / j["value"] < 0.01


def test_config_c4_and_c5_shard_the_fixed_job_over_the_ranks():
    """--config c4 / c5 (BASELINE configs 4 and 5): the job is fixed (1,048,576 channels / 4,096 receivers), N ranks take 1/N each
    (strong scaling), `value` counts the whole job's samples, `config.workload` names the config."""
    for cfg, total, blocks in (("c4", 1048576, 1), ("c5", 4096, 646)):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", cfg, "--dry-run", "--steps", "2", "--warmup", "1"],
                           capture_output=True, text=True, env=_env(), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["name"] == cfg and j["config"]["workload"].startswith(cfg.upper())
        assert j["config"]["channels_per_gpu"] == total // 2 and j["config"]["channels_total"] == total and j["config"]["blocks_per_step"] == blocks
        assert abs(j["value"] - total * blocks * 128 * 2 / (j["ms_per_step"] * 2e-3) / 1e6) / j["value"] < 0.01
        assert j["roofline"]["algorithmic_bytes_per_launch"] == int(round(j["roofline"]["algorithmic_bytes_per_channel_block"] * (total // 2) * blocks)) or \
            abs(j["roofline"]["algorithmic_bytes_per_launch"] / (total // 2 * blocks) - j["roofline"]["algorithmic_bytes_per_channel_block"]) < 0.1


def test_algorithmic_bytes_follow_survey_8d():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.ALGO_BYTES_PER_BLOCK == 10520 and bench.ALGO_READ_BYTES_PER_BLOCK == 5436
    assert abs(bench.C4_ALGO_BYTES_PER_BLOCK - 10501.14) < 0.01                                 # per-mode sums + ALS, averaged over c mod 7
    assert abs(bench.C5_ALGO_BYTES_PER_BLOCK - (768 * 646 + 96 + 2 * 1688) / 646.0) < 1e-9


def test_warmup_field_is_exactly_what_the_command_line_asked_for():
    """`warmup` in the JSON line = --warmup = the untimed steps of the measured batch (round 4: the clocks are settled on a scratch
    batch before the measured one exists; the dry run has no device to settle and says 0)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "3"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["warmup"] == 3 and j["steps"] == 2 and j["config"]["warmup_requested"] == 3
    assert j["config"]["clock_settle_launches_on_a_scratch_batch"] == 0


def test_gpus_1_dry_run_is_a_single_process_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    e = _env()
    e.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 2
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "refusing" in r.stderr


def test_cpu_info_reports_model_and_physical_cores():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    model, phys = bench.cpu_info()
    assert model is None or isinstance(model, str)
    assert phys is None or 1 <= phys <= (os.cpu_count() or 1)
