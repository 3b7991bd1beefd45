"""Host-side logic of bench.py that needs no GPU: the identity of the kernel sources a PMC profile belongs to and the rule
that a committed traffic figure is quoted only for those very sources and that very workload."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_kernel_source_sha_is_a_stable_hash_of_the_encode_kernels():
    a, b = bench.kernel_source_sha(), bench.kernel_source_sha()
    assert a == b and len(a) == 16 and int(a, 16) >= 0
    for f in bench.ENCODE_KERNEL_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "libzling_amd", "csrc", f)), f


def test_traffic_is_quoted_only_for_the_same_sources_and_workload(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"kernel_source_sha": "0" * 16, "workload": {"bytes": 1000, "level": 0, "blocks": 1},
             "kernels": {"k_rolz_parse_wave": {"hbm_bytes_corrected": 4242}}}
    (prof / "r09_z_pmc_traffic.json").write_text(json.dumps(entry))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "f" * 16)
    t, src = bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 1)
    assert t is None and src.startswith("stale")                      # other kernel sources
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "0" * 16)
    t, src = bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 1)
    assert t == 4242 and "r09_z_pmc_traffic.json" in src
    assert bench.traffic_of("k_rolz_parse_wave", 2000, 0, "synthetic", 1) == (None, None)     # another size
    assert bench.traffic_of("k_rolz_parse_wave", 1000, 4, "synthetic", 1) == (None, None)     # another level
    assert bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 2) == (None, None)     # several ranks


def test_the_committed_profile_names_its_kernel_sources():
    """Every committed PMC profile carries the hash of the sources it was taken on; a stale newest one is reported (bench.py
    then prints `traffic: null` with the reason) but is not an error of the code."""
    import warnings
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
    assert files
    newest = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    assert len(newest["kernel_source_sha"]) == 16
    if newest["kernel_source_sha"] != bench.kernel_source_sha():
        warnings.warn("profiles/%s was taken on other kernel sources: re-run scripts/profile_round.sh" % files[-1])


def test_schedule_model_arithmetic():
    """sharding.schedule_model / schedule_model_ranks (what bench.py prints as amdahl.model_ms) on hand-made stage times."""
    from libzling_amd import sharding
    # one context: parse, then rank + Huffman
    assert sharding.schedule_model([(500.0, 600.0, 2.0)]) == 1102.0
    # four equal contexts, all parses at once (rounds 1-3): one parse, then four finishes in a row
    st = [(3000.0, 1300.0, 3.0)] * 4
    assert sharding.schedule_model(st, 0) == 3000.0 + 4 * 1303.0
    # two parses in flight: parse ends 2000, 2000, 4000, 4000; finishes 3303, 4606, 5909, 7212
    st = [(2000.0, 1300.0, 3.0)] * 4
    assert sharding.schedule_model(st, 2) == 7212.0
    # a slow third parse makes its finish wait for it: parse ends 1000, 1000, 6000, 2000(+1000 start) ...
    st = [(1000.0, 100.0, 0.0), (1000.0, 100.0, 0.0), (5000.0, 100.0, 0.0), (1000.0, 100.0, 0.0)]
    assert sharding.schedule_model(st, 2) == 6000.0 + 100.0 + 100.0          # ctx 3 parsed long before, waits for ctx 2's finish
    # two ranks, one context each: both parse at once, rank 1's finish follows rank 0's
    assert sharding.schedule_model_ranks([[(600.0, 650.0, 2.0)], [(600.0, 650.0, 2.0)]], 2) == 600.0 + 2 * 652.0
    # ... and a rank whose parse is late holds its own finish up, not the earlier rank's
    assert sharding.schedule_model_ranks([[(100.0, 50.0, 0.0)], [(900.0, 50.0, 0.0)]], 2) == 950.0


def test_schedule_model_with_time_staggered_parses():
    """stagger = (first, gap_ms): context k is queued (k - first + 1) * gap_ms into the step."""
    from libzling_amd import sharding
    st = [(1900.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0)]
    # parses end 1900, 1200+2000 = 3200, 4400, 5600; finishes 3200, 4500, 5800, 7100
    assert sharding.schedule_model(st, 2, (1, 1200.0)) == 7100.0
    # a gap shorter than the chain's pace changes nothing once the chain is the bottleneck ...
    assert sharding.schedule_model(st, 2, (1, 1000.0)) == 7100.0
    # ... a longer one makes the last finish wait for its parse: 3 * 1500 + 2000 + 1300
    assert sharding.schedule_model(st, 2, (1, 1500.0)) == 7800.0
    # first = 2: two contexts at once, then one per gap
    assert sharding.schedule_model(st, 0, (2, 1200.0)) == max(2000.0, 1900.0 + 1300.0) + 1300.0 + 1300.0 + 1300.0


def test_schedule_model_with_explicit_launch_times():
    from libzling_amd import sharding
    st = [(1500.0, 600.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0)]
    # parses end 1500, 300 + 2000, 1300 + 2000; finishes 2100, 3600, 4900
    assert sharding.schedule_model(st, 2, ("at", [0.0, 300.0, 1300.0])) == 4900.0
