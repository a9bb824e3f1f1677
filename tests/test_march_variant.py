"""The column-march variant of the speculative forward (variants/src/scsfm_march.h; profiles/HISTORY.md 3a: measured slower than the tile
kernel, compiled only with -DSCSFM_WITH_MARCH, which the host-simulation build defines) stays parity-green: the
speculative-forward checks of test_hostsim_kernels.py re-run with SCSFM_SPEC_KERNEL=march (read per launch), at a
segment height that is no multiple of the chunk's, so that first / last chunks and carried rows are all exercised."""
import pytest

import test_hostsim_kernels as K


@pytest.fixture(scope="module")
def lib():
    from hostsim import harness
    return harness.lib()


@pytest.mark.parametrize("rows", ["24"])
def test_march_variant_matches_the_oracle(lib, monkeypatch, rows):
    monkeypatch.setenv("SCSFM_SPEC_KERNEL", "march")
    monkeypatch.setenv("SCSFM_MARCH_ROWS", rows)
    K.test_speculative_forward_fp64(lib, (0.7, 1.3), (0.7, 1.3))
    K.test_odd_image_sizes_with_open_gates(lib, 15, 63, 40, (1.0, 0.5))
    K.test_odd_image_sizes_with_open_gates(lib, 5, 200, 40, (1.0, 0.5))


def test_march_variant_fp32_goldens(lib, monkeypatch):
    monkeypatch.setenv("SCSFM_SPEC_KERNEL", "march")
    K.test_total_loss_fp32_matches_reference_goldens(lib, "smooth", False)
