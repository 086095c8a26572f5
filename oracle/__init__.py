"""CPU restatements of the reference's algorithms -- TEST INFRASTRUCTURE ONLY.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import this package; the product
(`lidar_transfer_amd/`) never does and fails loudly when its HIP library is missing.

  lt_oracle.c / binding.py   ray cast: the reference's BVH + traversal restated, brute force, LBVH model
                             (pinned against oracle/_ref = the reference's own C++ compiled in place, and the
                             golden vectors F2-F5)
  projection.py              do_range_projection / do_range_projection_new / label projection in numpy
                             (pinned against the reference's own arrays, golden vectors F6 and F9)
  lt_tsdf_oracle.c           TSDF `integrate` CUDA source restated in C; lt_tsdf_dense.hip the same for the GPU, bit-identical
                             to the reference's own kernel text compiled by hipcc for gfx950 (build_ref_tsdf.py +
                             ref_tsdf_launch.inc -> _ref/libref_tsdf_integrate{,_plain}.so); the plain-average branch
                             also against the reference's numpy CPU mode, golden F8
  lt_mc_oracle.c             get_mesh = scikit-image 0.18's Lewiner marching cubes + the attribute look-ups, restated in C:
                             returns the reference's ARRAYS (values and order) -- pinned by goldens F10 / F10b made with the
                             real scikit-image (tests/golden/make_golden_mc*.py, /opt/conda/bin/python3.9)
  gen_rsqrt_table.c          measures and exhaustively verifies the x86 RSQRTSS table the kernels replay
  Makefile                   builds liblt_oracle.so and oracle/_ref/ (needs /root/reference for the latter)
"""
