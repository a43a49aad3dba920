"""bench.py starts its own ranks when it is called as `python bench.py --gpus N` without a launcher (CPU test:
the ranks here are a stand-in script, started through the same torch.distributed.run command line)."""
import argparse
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "ranks.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_starts_n_ranks_and_relays_rank0s_line(tmp_path, capfd):
    script = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        print("banner of rank %d" % rank)
        assert os.environ["MASTER_ADDR"] == "127.0.0.1"
        if rank == 0:
            print(json.dumps({"metric": "Mrays/s (primary+secondary)", "n_gpus": world, "argv": sys.argv[1:]}))
    """)
    rc = bench.launch_ranks(argparse.Namespace(gpus=2), ["--gpus", "2", "--steps", "3"], script=script)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0
    assert len(out) == 1, out          # one JSON line, banners dropped
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["argv"] == ["--gpus", "2", "--steps", "3"]


def test_launcher_reports_a_failing_rank(tmp_path, capfd):
    script = _stub(tmp_path, """
        import os, sys
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
    """)
    rc = bench.launch_ranks(argparse.Namespace(gpus=2), ["--gpus", "2"], script=script)
    assert rc != 0


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_stale_counter_profiles_are_not_replayed(tmp_path, monkeypatch):
    """Counter files collected on other kernel sources give null fields and stale = true."""
    import raytracer_amd as rt
    prof = tmp_path / "profiles"
    prof.mkdir()
    pmc = {"workload": "temple 3840x2160", "csrc_sha256": "0" * 64,
           "k_shadow": {"SQ_INSTS_VALU": 3e9, "GRBM_GUI_ACTIVE": 8e7, "SQ_INSTS_SALU": 1e9}}
    (prof / "pmc.json").write_text(json.dumps(pmc))
    (prof / "traffic.json").write_text(json.dumps({"workload": "temple 3840x2160", "csrc_sha256": "0" * 64,
                                                   "k_shadow": {"hbm_bytes_per_launch": 1e9, "l2_hit_rate": 0.9, "l2_requests_per_launch": 1e8}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r = bench.roofline("shadow", 4.0, 1e6, 1e7, 1e9, 1e5, 1e8, "temple 3840x2160")
    assert r["stale"] is True and r["frac"] is None and r["traffic"] is None and r["hbm_frac"] is None
    assert r["logical_gbs"] > 0 and r["flops"] == 45 * 10 ** 5
    # the same files stamped with this tree's sources are used, and the fraction is one
    for name in ("pmc.json", "traffic.json"):
        j = json.loads((prof / name).read_text())
        j["csrc_sha256"] = rt.csrc_hash()
        (prof / name).write_text(json.dumps(j))
    r = bench.roofline("shadow", 4.0, 1e6, 1e7, 1e9, 1e5, 1e8, "temple 3840x2160")
    assert r["stale"] is False and 0.0 < r["frac"] <= 1.0 and 0.0 < r["hbm_frac"] < 1.0
    assert abs(r["frac"] - 2 * 3e9 / (1e7 * 1024)) < 1e-3
