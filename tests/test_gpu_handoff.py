"""The round-ending hand-off of the sumcheck-family kernels (zolt_amd/csrc/sc_common.hip.h: write-through partials, one relaxed arrival,
sc1 loads in the last arriver) under the conditions that expose a wrong one: every workgroup's L1 warmed with the previous launch's
partials, uneven arrival times, another stream streaming through HBM, every word checked by the device (zg_selftest_handoff).
The reference sums serially on one core (src/subprotocols/mod.zig:79-93); this exchange exists only here."""
import pytest

from zolt_amd import lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocks,threads", [
    (2, 64), (37, 256), (256, 256), (256, 512), (256, 1024),      # one counter (at most 256 arrivals)
    (257, 256), (512, 512), (1024, 256), (2048, 256), (2048, 64),  # two-level arrival, several workgroups per CU
])
@pytest.mark.parametrize("busy", [False, True])
def test_handoff_every_word_under_uneven_load(blocks, threads, busy):
    lib.init(0)
    iters = 300
    bad, done = lib.selftest_handoff(blocks, threads, iters, busy)
    assert done == iters, f"{done} of {iters} launches elected exactly one last arriver"
    assert bad == 0, f"{bad} stale or torn words read by the last arriver"


def test_handoff_rejects_bad_geometry():
    lib.init(0)
    for blocks, threads in [(1, 256), (4096, 256), (16, 100), (16, 2048)]:
        with pytest.raises(RuntimeError):
            lib.selftest_handoff(blocks, threads, 1, False)
