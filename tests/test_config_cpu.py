"""Sensor / approach YAML files are consumed as the reference reads them (lidar_deform.py:264-277, :302-315,
:318-355) and `bench.py --gpus N` becomes an N-rank job."""
import os
import sys

import numpy as np
import pytest

from lidar_transfer_amd.config import load_approach, load_sensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fname, H, W, up, down", [("hdl64_2048.yaml", 64, 2048, 3.0, -25.0),
                                                   ("hdl64_1024.yaml", 64, 1024, 3.0, -25.0),
                                                   ("vlp32_1024.yaml", 32, 1024, 10.0, -30.0),
                                                   ("os128_2048.yaml", 128, 2048, 15.0, -25.0)])
def test_sensor_yaml_shapes_of_the_baseline_configs(fname, H, W, up, down):
    s = load_sensor(os.path.join(ROOT, "config", fname))
    name, fov_up, fov_down, h, w, beam_angles = s.as_tuple()
    assert (h, w, fov_up, fov_down) == (H, W, up, down) and isinstance(name, str)
    rays = s.create_rays()
    assert rays.shape == (H * W, 3) and rays.dtype == np.float32


def test_width_is_truncated_float_division_and_beam_angles_are_sorted():
    # W = int(fov_hor / angle_res_hor) (lidar_deform.py:277): 360 / 0.2 = 1800, 360 / 0.7 = 514.28.. -> 514
    d = dict(name="x", fov_up=2, fov_down=-24.8, beams=64, angle_res_hor=0.7, fov_hor=360)
    assert load_sensor(d).W == 514 and load_sensor(d).beam_angles is None
    d["beam_angles"] = [3.0, -1.0, 2.0]
    assert load_sensor(d).beam_angles == [-1.0, 2.0, 3.0]
    assert d["beam_angles"] == [3.0, -1.0, 2.0]  # the caller's dict is not modified
    with pytest.raises(KeyError):
        load_sensor(dict(name="x", fov_up=2, fov_down=-24.8, beams=64))
    s = load_sensor(os.path.join(ROOT, "config", "vlp32_1024.yaml"))
    assert s.beam_angles == sorted(s.beam_angles) and len(s.beam_angles) == 32 and s.beam_angles[0] == -30.0


def test_approach_yaml():
    a = load_approach(dict(adaption="mergemesh", preserve_float=True, voxel_size=0.05, number_of_scans=5,
                           voxel_bounds=[-50, 50, -50, 50, -5, 5], ignore=[0, 1], moving=[252], batch_interval=10,
                           transformation=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
                           color_map={0: [0, 0, 0], 10: [245, 150, 100], 259: [255, 0, 0]}, labels={0: "unlabeled"}))
    assert a.voxel_bounds.shape == (3, 2) and a.voxel_bounds[2].tolist() == [-5, 5]
    lut = a.color_lut()
    assert lut.shape == (259 + 1 + 100, 3) and lut.dtype == np.float32  # laserscan.py:547-555
    assert np.array_equal(lut[10], np.array([245, 150, 100], np.float32) / 255.0)
    assert a.scan_indices(100) == list(range(2, 96, 10))  # nscans // 2 offset, stop before n - (nscans - 1)
    b = load_approach(dict(adaption="cp", preserve_float=False, voxel_size=0.1, number_of_scans=1, voxel_bounds=[0] * 6,
                           ignore=[], moving=[], transformation=[], color_map={}))
    assert b.batch_interval == 1  # lidar_deform.py:352-355


def test_bench_gpus_flag_builds_a_one_process_per_gpu_launch(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    # no GPUs here: asking for two must fail loudly instead of silently running one rank
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import torch
    if torch.cuda.device_count() < 2:
        with pytest.raises(SystemExit):
            bench.main()
