"""CPU: the tile/summary formulation the HIP step kernel implements (tests/tile_model.py) gives the
reference's a[]/d[] at every site, with tiny tiles so all cross-tile carries are exercised."""
import numpy as np
import pytest

from tile_model import step_tiles, summaries


@pytest.mark.parametrize("M,N,kind,T", [(37, 60, 1, 8), (100, 120, 0, 8), (64, 50, 1, 16), (257, 150, 0, 32), (1000, 60, 0, 64)])
def test_tile_model_matches_oracle(orc, M, N, kind, T):
    bits = orc.synth_bitcols(M, N, seed=M + N, kind=kind)
    hap = orc.unpack_bitcols(bits, M)
    o = orc.build_bitcols(bits, M, with_d=True, dump_sites=range(N + 1))
    a = np.arange(M)
    d = np.zeros(M + 1, np.int64)
    d[0] = d[M] = 1
    for k in range(N):
        assert np.array_equal(a, o["a_dump"][k]) and np.array_equal(d, o["d_dump"][k])
        y = hap[k][a]
        a, d = step_tiles(a, d, y, k, T, summaries(y, d, M, T))
    assert np.array_equal(a, o["a_dump"][N]) and np.array_equal(d, o["d_dump"][N])
