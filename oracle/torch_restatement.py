"""oracle/torch_restatement.py — second, independent restatement of the reference solver.

TEST INFRASTRUCTURE ONLY (see oracle/svnicp_oracle.h).  The reference (SVNICP.cpp / SVGDICP.cpp)
is a libtorch tensor program; this file replays it op-for-op with the SAME ATen operators through
the Python front end (torch CPU, float64), so that einsum/bmm/linalg_solve/linalg_inv/median run
the very kernels the reference would run on a CPU device.  It exists to cross-check the plain-C
oracle (svnicp_oracle.c) — the reference's own TUs cannot be compiled here (PCL/Eigen/GTSAM/rclcpp
absent; stand-in headers are not allowed), so solver parity is "unpinned" by the reference and
this is the closest available witness.  Slow and memory hungry (it materialises [P,B,3,6] like the
reference): small cases only.

Citations: file:line relative to /root/reference/svn-icp/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

F64 = torch.float64


@dataclass
class SteinICPParam:  # include/core/SVGDICP.h:41-57
    iterations: int = 50
    lr: float = 0.02
    max_dist: float = 1.0
    check_early_stop: bool = False
    convergence_threshold: float = 1e-5
    KNN_count: int = 100
    SVN_full_grad: bool = True
    optimizer: str = "Adam"


def knn_idx(p1: torch.Tensor, p2: torch.Tensor, K: int):
    """src/core/knn/knn_cpu.cpp:13-69 on float64: p1 [N,P1,3], p2 [N,P2,3] -> idx [N,P1,K], d2 [N,P1,K].
    K smallest by (dist, idx) with strict '<' insertion == stable ascending sort, first K."""
    N, P1, _ = p1.shape
    P2 = p2.shape[1]
    idx = torch.zeros((N, P1, K), dtype=torch.int64)
    dst = torch.zeros((N, P1, K), dtype=p1.dtype)
    step = max(1, (1 << 24) // max(P2, 1))
    for n in range(N):
        for s in range(0, P1, step):
            a = p1[n, s:s + step]
            dx = a[:, None, 0] - p2[n, None, :, 0]
            dy = a[:, None, 1] - p2[n, None, :, 1]
            dz = a[:, None, 2] - p2[n, None, :, 2]
            d = dx * dx            # dist = 0 + diff*diff, d = 0,1,2 (knn_cpu.cpp:43-50)
            d = d + dy * dy
            d = d + dz * dz
            sd, si = torch.sort(d, dim=1, stable=True)
            k = min(K, P2)
            idx[n, s:s + step, :k] = si[:, :k]
            dst[n, s:s + step, :k] = sd[:, :k]
    return idx, dst


class SVNICP:
    """svnicp::SVNICP (include/core/SVNICP.h:29-78, src/core/SVNICP.cpp)."""

    def __init__(self, param: SteinICPParam, init_pose: torch.Tensor):
        self.config = param
        self.K_source = param.KNN_count                       # SVGDICP.cpp:43
        self.R0 = torch.eye(3, dtype=F64)                     # SVGDICP.cpp:38
        self.t0 = torch.zeros((3, 1), dtype=F64)              # SVGDICP.cpp:39
        self._set_particles(init_pose)
        self.pose_particles = torch.cat(                      # SVNICP.cpp:36-37
            [self.t.view(self.P, 3).transpose(0, 1), self.rotm_to_ypr_tensor(self.R).transpose(0, 1)], 0)
        self.trace = None

    def _set_particles(self, init_pose):
        init_pose = init_pose.to(F64)
        self.P = P = init_pose.shape[1]
        self.x, self.y, self.z, self.rx, self.ry, self.rz = (init_pose[i].reshape(P, 1, 1) for i in range(6))
        self.R = self.to_rotation_tensor(self.rx, self.ry, self.rz)
        self.t = torch.cat([self.x, self.y, self.z], 1)       # SVGDICP.cpp:262-264

    def add_cloud(self, source, target, init_pose):           # SVGDICP.cpp:46-62
        self.source_cloud = source.to(F64).clone()
        self.target_cloud = target.to(F64).clone()
        self._set_particles(init_pose)

    def set_initial_mean(self, R0, t0):                       # SVGDICP.h:102-110
        self.R0 = torch.as_tensor(R0, dtype=F64).reshape(3, 3).clone()
        self.t0 = torch.as_tensor(t0, dtype=F64).reshape(3, 1).clone()

    # -- SVNICP.cpp:166-194
    def to_rotation_tensor(self, r, p, y):
        P = self.P
        zeros = torch.zeros(P, dtype=F64)
        angle_axis = torch.cat([r, p, y], 1).view(P, 3)
        angle = torch.norm(angle_axis, 2, 1).view(P, 1)
        zero_index = torch.lt(angle, 1e-12).view(P, 1)
        axis = torch.where(zero_index, torch.zeros((P, 3), dtype=F64), angle_axis.div(angle))
        cos_a, sin_a = torch.cos(angle), torch.sin(angle)
        a_hat = torch.stack([
            torch.stack([zeros, -axis[:, 2], axis[:, 1]], 1),
            torch.stack([axis[:, 2], zeros, -axis[:, 0]], 1),
            torch.stack([-axis[:, 1], axis[:, 0], zeros], 1)], 2).transpose(1, 2)
        eye = torch.eye(3, dtype=F64).expand(P, 3, 3)
        aaT = torch.matmul(axis.view(P, 3, 1), axis.view(P, 1, 3))
        R = torch.mul(cos_a.view(P, 1, 1), eye) + (1 - cos_a).view(P, 1, 1).mul(aaT) + sin_a.view(P, 1, 1).mul(a_hat)
        self.J_l = (torch.mul(sin_a.div(angle).view(P, 1, 1), eye)
                    + (1 - sin_a.div(angle)).view(P, 1, 1).mul(aaT)
                    + (1 - cos_a).div(angle).view(P, 1, 1).mul(a_hat))
        return R

    # -- SVNICP.cpp:196-215
    def rotm_to_ypr_tensor(self, R):
        P = self.P
        angle = torch.acos(torch.clip(0.5 * (R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1), -1, 1)).view(P, 1)
        sin_angle = torch.sin(angle)
        nonzero_mask = (sin_angle.abs() > 1e-12).view(P, 1)
        vee = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1).view(P, 3)
        angle_axis = 0.5 / sin_angle.view(P, 1).masked_fill(~nonzero_mask, 1) * angle * vee
        return torch.masked_fill(angle_axis, ~nonzero_mask, 0)

    # -- SVGDICP.cpp:176-215
    def mini_batch_pair_generator(self):
        Ns = self.source_cloud.shape[0]
        transformed = self.source_cloud.matmul(self.R0.transpose(0, 1)) + self.t0.view(1, 3)
        idx, d2 = knn_idx(transformed.view(1, Ns, 3), self.target_cloud.view(1, -1, 3), self.K_source)
        self.sourceKNN_idx = idx.reshape(Ns, self.K_source)
        self.sourceKNN_d2 = d2.reshape(Ns, self.K_source)
        target_batch = self.target_cloud[self.sourceKNN_idx.reshape(-1)].reshape(Ns, self.K_source, 3)
        return self.source_cloud, target_batch          # identical for every epoch (use_minibatch unset)

    # -- SVGDICP.cpp:300-333
    def get_correspondence_fast(self, source, transformed_source, target):
        P, B = self.P, source.shape[1]
        idx, d2 = knn_idx(transformed_source.transpose(0, 1).contiguous(), target, 1)
        dist_cr = d2.transpose(0, 1)                      # [P,B,1]
        target_index = idx.transpose(0, 1).reshape(P, B)
        paired = target.transpose(0, 1)[target_index, torch.arange(B)]
        m = torch.lt(dist_cr, self.config.max_dist / 1.0).to(F64)
        self.last_corr, self.last_mask = target_index, m.view(P, B)
        return m * source, m * transformed_source, m * paired

    # -- SVNICP.cpp:116-164
    def Newton_grad_right(self, source_paired, transformed_s_paired, target_paired):
        P, B = self.P, source_paired.shape[1]
        md = self.config.max_dist
        error = transformed_s_paired - target_paired
        error_squared = torch.norm(error, 2, 2, True)
        weight = torch.square(md / (md + 3 * error_squared))
        error = weight * error
        zero = torch.zeros((P, B), dtype=F64)
        sp = source_paired
        s_hat = torch.stack([
            torch.stack([zero, -sp[:, :, 2], sp[:, :, 1]], 2),
            torch.stack([sp[:, :, 2], zero, -sp[:, :, 0]], 2),
            torch.stack([-sp[:, :, 1], sp[:, :, 0], zero], 2)], 2)
        R_compound = self.R0.matmul(self.R).unsqueeze(1).expand(P, B, 3, 3)
        J = torch.cat([R_compound, -R_compound.matmul(s_hat)], 3)
        H = torch.einsum("pbik,pbil->pkl", J, J.mul(weight.view(P, B, 1, 1))) + 1e-6 * torch.eye(6, dtype=F64)
        b = torch.einsum("pbik, pbij->pk", J, error.view(P, B, 3, 1))
        Newton_grad = torch.linalg.solve(H, b)
        return Newton_grad, H, b

    # -- SVNICP.cpp:254-266
    def rbf_hessian_kernel(self, x1):
        P = self.P
        pair_difference = x1.view(P, 1, 6) - x1
        pairwise_square_norm = torch.mul(pair_difference, pair_difference).sum(2)
        h = torch.median(pairwise_square_norm) / math.log(x1.shape[0] + 1)
        return torch.exp(-pairwise_square_norm / h), h, pair_difference

    # -- SVNICP.cpp:218-227
    def svgd_grad(self, pose_parameters, newton_grad, H):
        P = self.P
        Kernel, bandwidth, pd = self.rbf_hessian_kernel(pose_parameters)
        grad = 2 / bandwidth * torch.mul(pd.view(P, P, 6), Kernel.view(P, P, 1)).sum(1)
        self.last_h = bandwidth
        return (Kernel.matmul(newton_grad) + torch.matmul(torch.linalg.inv(H), grad.view(P, 6, 1)).squeeze(2)) \
            / Kernel.sum(1, True)

    # -- SVNICP.cpp:229-252
    def svn_full_grad(self, pose_parameters, H, b):
        P = self.P
        Kernel, bandwidth, pd = self.rbf_hessian_kernel(pose_parameters)
        grad = 2 / bandwidth * torch.mul(pd.view(P, P, 6), Kernel.view(P, P, 1)).unsqueeze(-1)
        grad2 = torch.matmul(grad.view(P, P, 6, 1), grad.view(P, P, 1, 6)).sum(1)
        H_mean = ((Kernel.square().unsqueeze(-1).unsqueeze(-1) * H.unsqueeze(0)).sum(1) + grad2) / P
        svgd_update = (Kernel.matmul(b.squeeze()).view(P, 6, 1) + grad.sum(1)) / P
        self.last_h = bandwidth
        return self.config.lr * torch.linalg.inv(H_mean).matmul(svgd_update).squeeze()

    # -- SVNICP.cpp:268-279
    def pose_update(self, stein_grad):
        P = self.P
        d_R = self.to_rotation_tensor(stein_grad[:, 3].reshape(P, 1, 1), stein_grad[:, 4].reshape(P, 1, 1),
                                      stein_grad[:, 5].reshape(P, 1, 1))
        d_t = torch.cat([stein_grad[:, 0].reshape(P, 1, 1), stein_grad[:, 1].reshape(P, 1, 1),
                         stein_grad[:, 2].reshape(P, 1, 1)], 1)
        d_t = self.J_l.matmul(d_t)
        self.R = self.R.matmul(d_R)
        self.t = self.R.matmul(d_t) + self.t

    def _pose(self):
        return torch.cat([self.t.view(self.P, 3).transpose(0, 1), self.rotm_to_ypr_tensor(self.R).transpose(0, 1)], 0)

    # -- SVNICP.cpp:41-114
    def stein_align(self):
        P, cfg = self.P, self.config
        early_stop_threshold = torch.tensor([cfg.convergence_threshold])     # float32, 1-dim (:42)
        self.particle_weight = torch.ones((P, 1)) / P                        # float32 (:46)
        self.particle_stack = torch.zeros((cfg.iterations, 6, P))            # float32 (SVGDICP.cpp:173)
        source, target_batch = self.mini_batch_pair_generator()
        B = source.shape[0]
        tr = self.trace = dict(corr=[], mask=[], H=[], b=[], newton=[], phi=[], h=[], pose=[])
        for epoch in range(cfg.iterations):
            mini_batch_epoch = source.expand(P, B, 3)
            self.R_total = self.R0.matmul(self.R)
            self.t_total = self.t0 + self.R0.matmul(self.t)
            source_transformed = mini_batch_epoch.matmul(self.R_total.transpose(1, 2)) + self.t_total.view(P, 1, 3)
            sp, tsp, tp = self.get_correspondence_fast(mini_batch_epoch, source_transformed, target_batch)
            newton_grad, Hessian, b = self.Newton_grad_right(sp, tsp, tp)
            self.pose_particles = self._pose()
            self.last_h = torch.tensor(float("nan"), dtype=F64)
            if P > 1:
                if cfg.SVN_full_grad:
                    stein_grad = self.svn_full_grad(self.pose_particles.transpose(0, 1), Hessian, -b)
                else:
                    Hessian_mean = torch.mean(Hessian, 0).expand(P, 6, 6)
                    stein_grad = self.svgd_grad(self.pose_particles.transpose(0, 1), -newton_grad, Hessian_mean)
            else:
                stein_grad = -newton_grad.reshape(1, 6)
            tr["corr"].append(self.last_corr.clone()); tr["mask"].append(self.last_mask.clone())
            tr["H"].append(Hessian.clone()); tr["b"].append(b.clone()); tr["newton"].append(newton_grad.clone())
            tr["phi"].append(stein_grad.reshape(P, 6).clone()); tr["h"].append(float(self.last_h))
            self.pose_update(stein_grad.reshape(P, 6))
            if cfg.check_early_stop:
                if bool(torch.lt(stein_grad.reshape(P, 6).norm(2, 1).mean(0), early_stop_threshold)):
                    break
            self.pose_particles = self._pose()
            self.particle_stack[epoch] = self.pose_particles.view(6, P)
            tr["pose"].append(self.pose_particles.clone())
        self.pose_particles = self._pose()
        return 1  # ALIGN_SUCCESS

    # -- SVNICP.cpp:281-308, SVGDICP.cpp:515-534
    def get_transformation(self):
        return torch.mul(self.pose_particles, self.particle_weight.transpose(0, 1)).sum(1)

    def get_distribution(self):
        wm = self.get_transformation()
        return torch.mul((self.pose_particles - wm.view(6, 1)).square(), self.particle_weight.transpose(0, 1)).sum(1)

    def get_cov_matrix(self):
        P = self.P
        wm = self.get_transformation()
        diff = self.pose_particles - wm.view(6, 1)
        Sigma = torch.sum(self.particle_weight.view(P, 1, 1)
                          * torch.matmul(diff.transpose(0, 1).reshape(P, 6, 1), diff.transpose(0, 1).reshape(P, 1, 6)), 0)
        return Sigma.reshape(36)

    def get_particles(self):
        return self.pose_particles.reshape(-1)

    def get_particle_weight(self):
        return self.particle_weight.reshape(-1).to(F64)

    def get_particle_history(self):
        return self.particle_stack


class SVGDICP(SVNICP):
    """svnicp::SVGDICP first-order mode (src/core/SVGDICP.cpp:66-140, 226-260, 335-494)."""

    def __init__(self, param, init_pose):
        self.config = param
        self.K_source = param.KNN_count
        self.R0 = torch.eye(3, dtype=F64)
        self.t0 = torch.zeros((3, 1), dtype=F64)
        self._set_particles(init_pose)
        self.pose_particles = torch.stack([self.x, self.y, self.z, self.rx, self.ry, self.rz]).reshape(6, self.P)
        self.trace = None

    def _set_particles(self, init_pose):
        init_pose = init_pose.to(F64).clone()
        self.P = P = init_pose.shape[1]
        self.params = [init_pose[i].reshape(P, 1, 1).clone().requires_grad_(False) for i in range(6)]
        self.x, self.y, self.z, self.rx, self.ry, self.rz = self.params
        self.R = self.to_rotation_tensor(self.rx, self.ry, self.rz)
        self.t = torch.cat([self.x, self.y, self.z], 1)

    def to_rotation_tensor(self, r, p, y):                   # SVGDICP.cpp:226-260
        Cy, Sy, Cp, Sp, Cr, Sr = torch.cos(y), torch.sin(y), torch.cos(p), torch.sin(p), torch.cos(r), torch.sin(r)
        return torch.stack([
            torch.stack([Cp * Cy, Sr * Sp * Cy - Cr * Sy, Sr * Sy + Cr * Sp * Cy], 3).squeeze(1),
            torch.stack([Cp * Sy, Cr * Cy + Sr * Sp * Sy, Cr * Sp * Sy - Sr * Cy], 3).squeeze(1),
            torch.stack([-Sp, Sr * Cp, Cr * Cp], 3).squeeze(1)], 2).squeeze(1)

    def partial_derivative(self, roll, pitch, yaw):          # SVGDICP.cpp:335-396
        A, Bs, C, D, E, F = torch.cos(yaw), torch.sin(yaw), torch.cos(pitch), torch.sin(pitch), torch.cos(roll), torch.sin(roll)
        DE, DF, AC, AF, AE = D * E, D * F, A * C, A * F, A * E
        ADE, ADF, BC, BE, BF, BDE = A * DE, A * DF, Bs * C, Bs * E, Bs * F, Bs * DE
        z = torch.zeros_like(roll)

        def m(r0, r1, r2):
            return torch.stack([torch.stack(r0, 3).squeeze(1), torch.stack(r1, 3).squeeze(1),
                                torch.stack(r2, 3).squeeze(1)], 2).squeeze(1)
        pr = m([z, ADE + BF, BE - ADF], [z, -AF + BDE, Bs * (-DF) - AE], [z, C * E, C * (-F)])
        pp = m([A * -D, AC * F, AC * E], [Bs * -D, BC * F, BC * E], [-C, -DF, -DE])
        py = m([-BC, -Bs * DF - AE, AF - BDE], [AC, -BE + ADF, ADE + BF], [z, z, z])
        return torch.stack([self.R0.matmul(pr), self.R0.matmul(pp), self.R0.matmul(py)])

    def sgd_grad(self, sp, tsp, tp):                         # SVGDICP.cpp:398-455
        P = self.P
        md = self.config.max_dist
        pdv = self.partial_derivative(self.rx, self.ry, self.rz)
        nonzero_count = torch.count_nonzero(tsp.sum(2), dim=1).to(F64)
        error = tsp - tp
        error_squared = torch.norm(error, 2, 2, True)
        error = torch.square(md / (md + 3 * error_squared)) * error
        g = torch.zeros((P, 6), dtype=F64)
        g[:, 0:3] = error.sum(1).matmul(self.R0) / (nonzero_count + 1).reshape(P, 1)
        for a in range(3):
            g[:, 3 + a] = torch.einsum("pbc, pbc->pb", error, torch.einsum("prc, pbc->pbr", pdv[a], sp)).sum(1) \
                / (nonzero_count + 1)
        return g * float(self.source_cloud.shape[0])

    def svgd_grad_first_order(self, pose_parameters, sgd_grad):   # SVGDICP.cpp:457-474
        P = self.P
        Kernel, bandwidth, pd = self.rbf_hessian_kernel(pose_parameters)
        grad = 2 / bandwidth * torch.mul(pd.view(P, P, 6), Kernel.view(P, P, 1)).sum(1)
        self.last_h = bandwidth
        return (Kernel.matmul(sgd_grad) + grad) / pose_parameters.shape[0]

    def _make_optimizer(self):                               # SVGDICP.cpp:142-170
        lr, name = self.config.lr, self.config.optimizer
        ps = [p.requires_grad_(True) for p in self.params]
        if name == "Adam":
            return torch.optim.Adam(ps, lr=lr, betas=(0.9, 0.999))
        if name == "RMSprop":
            return torch.optim.RMSprop(ps, lr=lr, weight_decay=1e-8, momentum=0.9)
        if name == "SGD":
            return torch.optim.SGD(ps, lr=lr)
        if name == "Adagrad":
            return torch.optim.Adagrad(ps, lr=lr)
        return None

    def _pose(self):
        return torch.stack([p.detach() for p in self.params]).reshape(6, self.P).clone()

    def stein_align(self):                                   # SVGDICP.cpp:66-140
        P, cfg = self.P, self.config
        opt = self._make_optimizer()
        if opt is None:
            return 2  # NO_OPTIMIZER
        self.particle_stack = torch.zeros((cfg.iterations, 6, P))
        source, target_batch = self.mini_batch_pair_generator()
        B = source.shape[0]
        tr = self.trace = dict(corr=[], mask=[], newton=[], phi=[], h=[], pose=[])
        thr = torch.tensor([cfg.convergence_threshold])
        for epoch in range(cfg.iterations):
            with torch.no_grad():
                mini_batch_epoch = source.expand(P, B, 3)
                self.R = self.to_rotation_tensor(self.rx, self.ry, self.rz)
                self.t = torch.cat([self.x, self.y, self.z], 1)
                self.R_total = self.R0.matmul(self.R)
                self.t_total = self.t0 + self.R0.matmul(self.t)
                source_transformed = mini_batch_epoch.matmul(self.R_total.transpose(1, 2)) + self.t_total.view(P, 1, 3)
                sp, tsp, tp = self.get_correspondence_fast(mini_batch_epoch, source_transformed, target_batch)
                g = self.sgd_grad(sp, tsp, tp)
                self.last_h = torch.tensor(float("nan"), dtype=F64)
                if P > 1:
                    stein_grad = self.svgd_grad_first_order(self.pose_particles.transpose(0, 1), -g)
                else:
                    stein_grad = -g
                tr["corr"].append(self.last_corr.clone()); tr["mask"].append(self.last_mask.clone())
                tr["newton"].append(g.clone()); tr["phi"].append(stein_grad.clone()); tr["h"].append(float(self.last_h))
                old = self.pose_particles.clone()
                for i, p in enumerate(self.params):          # SVGDICP.cpp:476-494
                    p.grad = -stein_grad[:, i].reshape(P, 1, 1).clone()
            opt.step()
            opt.zero_grad()
            with torch.no_grad():
                self.pose_particles = self._pose()
                diff = self.pose_particles - old
                if cfg.check_early_stop and bool(torch.lt(diff.norm(2, 0).mean(0), thr)):
                    break
                self.particle_stack[epoch] = self.pose_particles.view(6, P)
                tr["pose"].append(self.pose_particles.clone())
        self.pose_particles = self._pose()
        return 1

    def get_transformation(self):                            # SVGDICP.cpp:497-499
        return torch.mean(self.pose_particles, 1)

    def get_distribution(self):                              # SVGDICP.cpp:501-503
        return torch.var(self.pose_particles, 1)

    def get_cov_matrix(self):                                # SVGDICP.cpp:505-513
        P = self.P
        diff = self.pose_particles - self.get_transformation().view(6, 1)
        return torch.mean(torch.matmul(diff.transpose(0, 1).reshape(P, 6, 1), diff.transpose(0, 1).reshape(P, 1, 6)), 0).reshape(36)

    def get_particle_weight(self):                           # SVGDICP.cpp:522-524
        return torch.ones(self.P, dtype=F64)
