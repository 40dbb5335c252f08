"""Caller glue on the GPU box: a short synthetic drive through the scene, scan-to-map with the HIP solver."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_registration_pipeline_tracks_a_synthetic_drive(hip):
    import importlib
    pl = importlib.import_module(hip.__name__ + ".pipeline")
    sm = importlib.import_module(hip.__name__ + ".stein_msgs")
    sc = hip.scans
    scene = sc.make_scene()
    cfg = pl.PipelineConfig(min_range=1.0, max_range=80.0, voxel_size=0.5, map_voxel_size=0.5, map_voxel_max_points=20,
                            map_range=100.0, particle_count=32,
                            solver=hip.SteinICPParam(iterations=30, lr=1.0, max_dist=1.0, KNN_count=50))
    pipe = pl.RegistrationPipeline(cfg, device=0)
    truth, est, guess = [], [], []
    for k in range(8):
        # climb + yaw: the ground plane and the walls observe both well (sliding along them they do not — with
        # point-to-point residuals every plane correspondence also pins the in-plane position)
        t = np.array([0.0, 0.0, 0.05 * k])
        R = sc.rot_zyx(0.0, 0.0, np.radians(0.3 * k))
        pts = sc.lidar_scan(scene, R, t, 32768, stream=300 + k)
        res = pipe.process_scan(pts, stamp=0.1 * k)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        truth.append(T); est.append(res.pose); guess.append(res.initial_guess)
        assert np.isfinite(res.pose).all()
        if k >= 1:
            assert res.state == int(hip.SteinICPState.ALIGN_SUCCESS)
            assert res.particles.shape == (6 * 32,) and abs(res.weights.sum() - 1.0) < 1e-6
            assert np.allclose(res.pose, res.initial_guess @ pl.correction_to_pose(res.correction))   # updater_, :37-46
            msg = sm.decode("stein_msgs/SteinParticle", sm.encode(sm.fill_particle(res.particles, res.weights, res.stamp)))
            assert msg["x"] == res.particles[:32].tolist()
    # The first frame defines the map frame; afterwards the estimate follows the planted climb and yaw.
    ez_guess = np.array([abs(T[2, 3] - G[2, 3]) for T, G in zip(truth, guess)])
    ez = np.array([abs(T[2, 3] - E[2, 3]) for T, E in zip(truth, est)])
    eyaw = np.array([np.linalg.norm(pl.so3_log(T[:3, :3].T @ E[:3, :3])) for T, E in zip(truth, est)])
    print("z error of prediction / estimate per frame:", np.round(ez_guess, 3), np.round(ez, 3), "rot err", np.round(eyaw, 4))
    assert ez[1] < 0.5 * ez_guess[1]                      # first registration: the prediction is a full step off
    assert ez.max() < 0.03 and est[-1][2, 3] > 0.9 * truth[-1][2, 3]
    assert eyaw[-1] < 0.01
    assert len(pipe.map) > 1000
