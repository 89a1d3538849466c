"""GPU (-m gpu): a fixed budget of randomised differential cases (tests/fuzz_cases.py): widths around every tile boundary,
random batch sizes and pass splits, both panel kinds; device pass API, packed consumers, host build + read side, record
sinks, dense and sparse query sweeps — each against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,ncases", [(1, 400), (2, 400), (20260928, 400)])
def test_fuzz_against_oracle(gpu_lib, orc, seed, ncases):
    from fuzz_cases import run_cases
    n, bad = run_cases(seed, ncases=ncases)
    assert n == ncases and not bad, "\n".join(bad)
