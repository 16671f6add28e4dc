"""CPU: no committed profile of the CURRENT round may be stale ("make every profile refuse to go stale").

Every file under profiles/ that DESIGN.md quotes for the tree at HEAD carries a stamp written by the tool that produced it on the GPU
box: the content hashes of the device sources its kernels come from (bench.stamp). This test recomputes those hashes from the tree and
fails when the newest rNN_* profile of a kind was measured on other sources - the file must then be re-collected (tools/roundend.sh,
tools/pmc_traffic.sh, tools/pmc_util.sh, tools/membound_prof.sh) or the kernel change reverted. Profiles of earlier rounds are history
and are not checked. A CSV (rocprofv3's own output) has a sidecar `<name>.stamp.json`."""
import json
import re
from pathlib import Path

import pytest

import bench

PROFILES = Path(__file__).resolve().parent.parent / "profiles"


def newest_round():
    rounds = sorted({m.group(1) for p in PROFILES.glob("r[0-9][0-9]_*") if (m := re.match(r"(r\d\d)_", p.name))})
    return rounds[-1]


def newest(pattern):
    """The newest file of the newest round matching `pattern` (with {r} = the round), by version suffix _vN if any."""
    r = newest_round()
    files = sorted(PROFILES.glob(pattern.format(r=r)), key=lambda p: [int(x) for x in re.findall(r"_v(\d+)", p.name)] or [0])
    return files[-1] if files else None


KINDS = [
    ("{r}_pmc_traffic.json", bench.BENCH_SOURCES),
    ("{r}_pmc_util.json", bench.BENCH_SOURCES),
    ("{r}_bench_kernel_stats*.csv", bench.BENCH_SOURCES),
    ("{r}_membound_rocprof.json", bench.MEMBOUND_SOURCES),
    ("{r}_membound_kernel_stats.csv", bench.MEMBOUND_SOURCES),
]


@pytest.mark.parametrize("pattern,sources", KINDS)
def test_newest_profile_was_measured_on_these_sources(pattern, sources):
    if newest_round() < "r04":
        pytest.skip("the stamps start with round 4; earlier rounds' profiles are history")
    f = newest(pattern)
    if f is None:
        pytest.skip(f"no {pattern} in round {newest_round()} yet")
    if f.suffix == ".csv":
        side = f.with_name(f.stem + ".stamp.json")
        assert side.exists(), f"{f.name} has no sidecar stamp {side.name}: re-collect it with the round's tools"
        obj = json.loads(side.read_text())
    else:
        obj = json.loads(f.read_text())
    assert "device_src_files" in obj or "device_src_sha" in obj, f"{f.name} carries no source stamp"
    stale = {n: (obj.get("device_src_files", {}).get(n), h) for n, h in bench.device_src_shas(sources).items()
             if obj.get("device_src_files", {}).get(n) != h}
    assert bench.stamp_is_current(obj, sources), f"{f.name} was measured on other device sources {stale}: re-collect it before quoting it"


def test_stamp_logic():
    now = bench.stamp(bench.BENCH_SOURCES)
    assert set(now["device_src_files"]) == set(bench.BENCH_SOURCES) and bench.stamp_is_current(now, bench.BENCH_SOURCES)
    bad = json.loads(json.dumps(now))
    bad["device_src_files"]["gemm.hip"] = "0" * 16
    assert not bench.stamp_is_current(bad, bench.BENCH_SOURCES)
    assert bench.stamp_is_current({"device_src_sha": bench.device_src_sha()}, bench.BENCH_SOURCES)  # a round-3 stamp: tree-wide
    assert not bench.stamp_is_current({"device_src_sha": "x"}, bench.BENCH_SOURCES) and not bench.stamp_is_current({}, bench.BENCH_SOURCES)
