"""lidar_transfer_amd -- the virtual-LiDAR ray-cast path of PRBonn/lidar_transfer on MI355X (gfx950).

Host-side mirror of the reference's interfaces for that path; every function ends in ``liblidarhip.so``
(``include/lidarhip.h``, hand-written HIP in ``csrc/``) and raises if the library is missing -- there is no CPU
implementation in this package (``oracle/`` is test infrastructure).

    raytracer   C_Trace (drop-in for RayTracerCython.C_Trace), Scene / RaySet (device-resident API)
    fusion      throw_rays_at_mesh, TSDFVolume (fusion_lidar.py)
    laserscan   create_rays, LaserScan / SemLaserScan projections (laserscan.py)
    post        do_reverse_projection_new, pack_scan / write_scan, compare (laserscan.py, np_ioueval.py)
    pipeline    ScanPipeline: the batch loop body, batches of scans in flight on one GPU; HostScanPipeline (host meshes
                over PCIe); FusionScanPipeline (fuse -> marching cubes -> render per output scan, chains in flight)
    dist        scan_indices / partition / render_scans / gather_to_root: one process per GPU, one gather
    synth       synthetic scenes and workloads of SURVEY.md section 8d
    build       hipcc build of liblidarhip.so (in-tree)
"""
