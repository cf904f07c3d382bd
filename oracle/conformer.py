"""Oracle restatement of the pose-update arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  axis_angle_to_matrix          utils/geometry.py:7-86 (quaternion route, small-angle Taylor)
  kabsch_batch                  utils/geometry.py:246-276 (rigid_transform_Kabsch_3D_torch_batch)
  torsion_update_batch          utils/torsion.py:75-90
  modify_conformer_batch        utils/diffusion_utils.py:60-78
  t_to_sigma / t_schedule / sinusoidal_embedding
                                utils/diffusion_utils.py:28-32,138-143,99-110
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def axis_angle_to_matrix(aa):
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * s], -1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def kabsch_batch(A, B):
    """A, B: [b, N, 3].  Returns R [b,3,3], t [b,3,1] with  A @ R^T + t^T ~= B."""
    A, B = A.permute(0, 2, 1), B.permute(0, 2, 1)
    cA, cB = A.mean(2, keepdim=True), B.mean(2, keepdim=True)
    H = torch.bmm(A - cA, (B - cB).transpose(1, 2))
    U, S, Vt = torch.linalg.svd(H)
    R = torch.bmm(Vt.transpose(1, 2), U.transpose(1, 2))
    SS = torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=A.dtype))
    Rm = torch.bmm(Vt.transpose(1, 2) @ SS, U.transpose(1, 2))
    R = torch.where(torch.linalg.det(R)[:, None, None] < 0, Rm, R)
    t = torch.bmm(-R, cA) + cB
    return R, t


def torsion_update_batch(pos, rot_edges, mask_rotate, torsion_updates):
    """pos [b,N,3]; rot_edges [R,2] (u,v); mask_rotate bool [R,N]; torsion_updates [b,R].
    Sequential over bonds in edge order; rotates mask atoms about (pos[u]-pos[v])."""
    pos = pos.clone()
    for idx in range(rot_edges.shape[0]):
        u, v = int(rot_edges[idx, 0]), int(rot_edges[idx, 1])
        assert not bool(mask_rotate[idx, u]) and bool(mask_rotate[idx, v])
        vec = pos[:, u] - pos[:, v]
        rot = axis_angle_to_matrix(vec / torch.linalg.norm(vec, dim=-1, keepdim=True) * torsion_updates[:, idx:idx + 1])
        m = mask_rotate[idx]
        pos[:, m] = torch.bmm(pos[:, m] - pos[:, v:v + 1], rot.transpose(1, 2)) + pos[:, v:v + 1]
    return pos


def modify_conformer_batch(orig_pos, B, rot_edges, mask_rotate, tr_update, rot_update, torsion_updates):
    """orig_pos [B*N,3] -> new [B*N,3].  rot_edges: the rotatable directed bonds (u,v) of ONE
    sample, local atom indices, in edge order (edge_index[:, :M].T[edge_mask[:M]])."""
    N = orig_pos.shape[0] // B
    pos = orig_pos.reshape(B, N, 3) + 0
    center = pos.mean(1, keepdim=True)
    rot = axis_angle_to_matrix(rot_update)
    rigid = torch.bmm(pos - center, rot.permute(0, 2, 1)) + tr_update.unsqueeze(1) + center
    if torsion_updates is None:
        return rigid.reshape(-1, 3)
    flex = torsion_update_batch(rigid, rot_edges, mask_rotate, torsion_updates.reshape(B, -1))
    R, t = kabsch_batch(flex, rigid)
    return (torch.bmm(flex, R.transpose(1, 2)) + t.transpose(1, 2)).reshape(-1, 3)


def t_to_sigma(cfg, t_tr, t_rot, t_tor):
    return (cfg.tr_sigma_min ** (1 - t_tr) * cfg.tr_sigma_max ** t_tr,
            cfg.rot_sigma_min ** (1 - t_rot) * cfg.rot_sigma_max ** t_rot,
            cfg.tor_sigma_min ** (1 - t_tor) * cfg.tor_sigma_max ** t_tor)


def get_t_schedule(inference_steps, t_max=1.0):
    """'expbeta' with alpha=beta=1: beta.cdf / beta.ppf are the identity on [0,1]."""
    return np.linspace(t_max, 0, inference_steps + 1)[:-1]


def sinusoidal_embedding(timesteps, embedding_dim, max_positions=10000):
    half = embedding_dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], 1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb
