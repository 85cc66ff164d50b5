"""`python bench.py --gpus N` must be launchable exactly as the driver does it (VERDICT r1 missing #1): with WORLD_SIZE unset it
spawns N ranks itself, with WORLD_SIZE set (torch.distributed.run) each rank runs. CPU-only: the --plumbing-only leg does the
rendezvous, builds both shard layouts and round-trips the frame<->pixel all-to-alls over gloo without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_gpus_2_spawns_two_ranks_by_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = _last_json(r.stdout)
    assert res["n_gpus"] == 2 and res["plumbing_only"] is True
    assert res["layouts"]["hybrid"]["t_counts"] == [25] and res["layouts"]["hybrid"]["cfg_half"] == 0   # 2 ranks = pure CFG split
    assert res["layouts"]["frames"]["t_counts"] == [13, 12]


def test_bench_under_torchrun_eight_ranks_config3_layout():
    """The driver's own launch line for N > 1 (torch.distributed.run sets WORLD_SIZE): 8 ranks, BASELINE config 3's 4/3/3/3/3/3/3/3."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
           "29631", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-only"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = _last_json(r.stdout)
    assert res["n_gpus"] == 8
    assert res["layouts"]["frames"]["t_counts"] == [4, 3, 3, 3, 3, 3, 3, 3]
    assert res["layouts"]["hybrid"]["t_counts"] == [7, 6, 6, 6]


def test_bench_rejects_mismatched_world():
    env = _clean_env()
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_roofline_traffic_is_tied_to_the_kernel_source(tmp_path):
    """bench.py reports `roofline.traffic` only from a PMC record stamped with the sha256 of the csrc/attention.hip that is in the
    tree; the tracked record must match the tracked source (a kernel edit without a re-measurement / re-stamp fails HERE, not silently in the
    driver's line), and a foreign record yields None."""
    import hashlib
    import json
    import shutil
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traffic, src = bench.attn_traffic_record(root)
    assert isinstance(traffic, int) and 1.0e9 < traffic < 3.0e9, src       # 1.18 GB algorithmic; measured 1.5-1.6 GB
    fake = tmp_path / "repo"
    (fake / "vista_amd" / "csrc").mkdir(parents=True)
    (fake / "profiles").mkdir()
    shutil.copy(os.path.join(root, "vista_amd", "csrc", "attention.hip"), fake / "vista_amd" / "csrc" / "attention.hip")
    with open(fake / "vista_amd" / "csrc" / "attention.hip", "a") as f:
        f.write("// edited\n")
    rec = {"attention_hip_sha256": hashlib.sha256(open(os.path.join(root, "vista_amd", "csrc", "attention.hip"), "rb").read()).hexdigest(), "traffic_bytes_per_launch": 123}
    json.dump(rec, open(fake / "profiles" / "r09_attn_traffic.json", "w"))
    t2, src2 = bench.attn_traffic_record(str(fake))
    assert t2 is None and "mismatch" in src2
