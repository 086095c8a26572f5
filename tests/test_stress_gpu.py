"""Randomised comparisons with the two checkers that ARE the reference's code run here -- short versions of the stress runs of
tools/r04/r04_final.sh (tests/stress_mc.py 300 volumes, tests/stress_tsdf_ref.py 300 configurations; profiles/r04/gpu_suite.txt)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_marching_cubes_on_random_volumes_equals_scikit_images_arrays(oracle):
    import stress_mc
    assert stress_mc.main(["stress_mc", "60", "11"]) == 0


def test_integrate_on_random_configurations_equals_the_reference_kernel_build():
    from oracle import binding as ob
    if not ob.ref_tsdf_available():
        pytest.skip("oracle/_ref/libref_tsdf_integrate.so not built (needs /root/reference + hipcc at build time)")
    import stress_tsdf_ref
    assert stress_tsdf_ref.main(["stress_tsdf_ref", "60", "11"]) == 0
