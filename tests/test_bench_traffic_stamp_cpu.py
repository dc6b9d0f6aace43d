"""CPU: `roofline.traffic` in bench.py's JSON line comes from profiles/pmc_traffic.json (rocprofv3 --pmc passes, not collected in the run).  The file is
stamped with a digest of the kernel sources + build flags and bench.py quotes it only when that digest matches the sources it runs -- a counter set
collected on older kernels must never describe newer ones."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_traffic_matches_the_committed_kernel_sources():
    from pmc_traffic import csrc_digest
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert t.get("csrc_digest"), "profiles/pmc_traffic.json carries no csrc digest"
    assert t["csrc_digest"] == csrc_digest(), ("profiles/pmc_traffic.json was collected on other kernel sources than the ones in csrc/: re-run "
                                               "tools/sessions/closing.sh (bench.py would report traffic = null)")
    assert t["igemm_bytes_per_launch"] > 1e8 and t["attention_bytes_per_launch"] > 1e8


def test_digest_follows_the_sources(tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("ada_build_t", os.path.join(ROOT, "amodal-depth-anything_amd", "csrc", "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    d0 = B._digest([])
    assert d0 == B._digest([]) and d0 != B._digest(["-DADA_OPERAND_BF16"])
