"""`sampling()` with the reference's signature (utils/sampling.py:69-72), running the step loop on the device.

    data_list, confidence = sampling(data_list, model, inference_steps, tr_schedule, rot_schedule,
                                     tor_schedule, device, t_to_sigma, model_args, ...)

Differences, all documented: the Gaussian draws come from a counter-based generator keyed by
(seed, global sample index, step, component) instead of the global torch RNG (so that a run sharded over GPUs
reproduces the single-GPU trajectories), or are injected through `noise=`; the confidence model is called after each
batch exactly as in the reference (sampling.py:208-227: fresh ligand positions copied into the confidence graphs,
t = 0) when it is a confidence-mode model of the built classes; visualisation hooks / full trajectories / feature
returns are not on the built path and raise NotImplementedError.  Per-step `crop_beyond` (sampling.py:104-109) runs on the device as a residue
mask + contact-graph re-compaction instead of the reference's deepcopy / to_data_list / from_data_list round trip.
"""
from __future__ import annotations

import numpy as np
import torch

from .hetero import HeteroBatch, set_time


def _batches(data_list, batch_size):
    for lo in range(0, len(data_list), batch_size):
        yield lo, data_list[lo:lo + batch_size]


def _collate(chunk):
    try:  # real PyG objects collate themselves
        from torch_geometric.data import Batch  # type: ignore
        return Batch.from_data_list(chunk)
    except Exception:
        return HeteroBatch.from_data_list(chunk)


def _edge_store(g, a, b):
    """Edge store of node types (a, b) whatever the relation name (('receptor', 'rec_contact', 'receptor') in the reference)."""
    for et in g.edge_types:
        if et[0] == a and et[-1] == b:
            return g[et]
    return g[a, b]


def crop_beyond(graph, cutoff, all_atoms=False):
    """utils/utils.py:388-413 on ONE complex graph, in place: residues (and, all_atoms, their atoms) farther than `cutoff`
    from every ligand atom are removed, contact graphs restricted to the kept nodes and relabelled.  Host tensors: the
    reference applies it to the confidence graphs once per batch (utils/sampling.py:213-217); the per-step crop of the score
    model runs on the device instead (ddmi_set_crop_cutoff)."""
    lig, rec = graph["ligand"].pos, graph["receptor"].pos
    keep = torch.any(torch.sum((lig.unsqueeze(0) - rec.unsqueeze(1)) ** 2, -1) < cutoff ** 2, dim=1)
    if not bool(keep.any()):
        raise ValueError(f"crop_beyond({cutoff}) removes every residue of '{getattr(graph, 'name', '?')}': the ligand is farther than "
                         f"the cutoff from the whole receptor (an empty receptor graph cannot be scored)")

    def sub_graph(mask, edge_index):
        ok = mask[edge_index[0]] & mask[edge_index[1]]
        return (torch.cumsum(mask.long(), 0) - 1)[edge_index[:, ok]]
    if all_atoms:
        ar = _edge_store(graph, "atom", "receptor")
        a2r = ar.edge_index[1]
        atoms_keep = keep[a2r]
        new_a2r = (torch.cumsum(keep.long(), 0) - 1)[a2r][atoms_keep]
    graph["receptor"].pos = rec[keep]
    graph["receptor"].x = graph["receptor"].x[keep]
    if hasattr(graph["receptor"], "side_chain_vecs") and graph["receptor"].side_chain_vecs is not None:
        graph["receptor"].side_chain_vecs = graph["receptor"].side_chain_vecs[keep]
    rr = _edge_store(graph, "receptor", "receptor")
    rr.edge_index = sub_graph(keep, rr.edge_index)
    if all_atoms:
        graph["atom"].x = graph["atom"].x[atoms_keep]
        graph["atom"].pos = graph["atom"].pos[atoms_keep]
        aa = _edge_store(graph, "atom", "atom")
        aa.edge_index = sub_graph(atoms_keep, aa.edge_index)
        ar.edge_index = torch.stack([torch.arange(len(new_a2r), device=new_a2r.device), new_a2r])
    return graph


def step_coefficients(model_args, t_idx, inference_steps, schedules, ode=False, no_random=False, no_final_step_noise=False,
                      temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5):
    """Host float64 scalars of one step (utils/sampling.py:97-186): per component (score coefficient, noise
    coefficient).  Mirrors the in-library computation of ddmi_sample; used by the step-wise python loop."""
    three = lambda v: list(v) if hasattr(v, "__iter__") else [v] * 3
    T, psi, sdat = three(temp_sampling), three(temp_psi), three(temp_sigma_data)
    out = []
    last = t_idx == inference_steps - 1
    for i, (name, sched) in enumerate(zip(("tr", "rot", "tor"), schedules)):
        t = float(sched[t_idx])
        dt = t if last else t - float(sched[t_idx + 1])
        smin, smax = getattr(model_args, f"{name}_sigma_min"), getattr(model_args, f"{name}_sigma_max")
        sigma = smin ** (1 - t) * smax ** t
        g = sigma * np.sqrt(2 * np.log(smax / smin))
        a, z = (0.5 * g * g * dt if ode else g * g * dt), g * np.sqrt(dt)
        if T[i] != 1.0:
            sigma_data = np.exp(sdat[i] * np.log(smax) + (1 - sdat[i]) * np.log(smin))
            lam = (sigma_data + sigma) / (sigma_data + sigma / T[i])
            a, z = g * g * dt * (lam + T[i] * psi[i] / 2), g * np.sqrt(dt * (1 + psi[i]))
        if no_random or ode or (no_final_step_noise and last):
            z = 0.0
        out.append((float(a), float(z)))
    return out


def sampling(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device=None, t_to_sigma=None,
             model_args=None, no_random=False, ode=False, visualization_list=None, confidence_model=None,
             confidence_data_list=None, confidence_model_args=None, t_schedule=None, batch_size=32,
             no_final_step_noise=False, pivot=None, return_full_trajectory=False, temp_sampling=1.0, temp_psi=0.0,
             temp_sigma_data=0.5, return_features=False, seed=0, noise=None, sample_id_offset=0, native_loop=True):
    if visualization_list is not None or return_full_trajectory or return_features or pivot:
        raise NotImplementedError("visualisation / trajectories / feature returns are outside the built path")
    confidence = [] if confidence_model is not None else None
    conf_batches = None
    conf_crop = getattr(confidence_model_args, "crop_beyond", None) if confidence_model_args is not None else None
    if confidence_model is not None and confidence_data_list is not None:
        conf_batches = iter([c for _, c in _batches(confidence_data_list, batch_size)])   # DataLoader order, sampling.py:87
    crop = getattr(model_args, "crop_beyond", None) if model_args is not None else getattr(model.cfg, "crop_beyond", None)
    N = len(data_list)
    schedules = (np.asarray(tr_schedule, dtype=np.float64), np.asarray(rot_schedule, dtype=np.float64),
                 np.asarray(tor_schedule, dtype=np.float64))
    cfg = model.cfg if model_args is None else model_args
    with torch.no_grad():
        for lo, chunk in _batches(data_list, batch_size):
            batch = _collate(chunk)
            b = batch.num_graphs
            n = batch["ligand"].pos.shape[0] // b
            if device is not None:
                batch = batch.to(device)
            ids = list(range(sample_id_offset + lo, sample_id_offset + lo + b))
            R = int(batch["ligand"].edge_mask.sum()) // b
            z = None
            if noise is not None:
                z = (noise[0][:, lo:lo + b], noise[1][:, lo:lo + b], noise[2][:, lo * R:(lo + b) * R])
            if native_loop and hasattr(model, "sample_batch"):
                pos = model.sample_batch(batch, inference_steps, schedules, noise=z, seed=seed, sample_ids=ids, ode=ode,
                                         no_random=no_random, no_final_step_noise=no_final_step_noise,
                                         temp_sampling=temp_sampling, temp_psi=temp_psi, temp_sigma_data=temp_sigma_data,
                                         crop_beyond=crop)
            else:   # step-wise: model(batch) per step, exactly the reference's loop structure
                pos = batch["ligand"].pos
                try:
                    for t_idx in range(inference_steps):
                        set_time(batch, schedules[0][t_idx], schedules[1][t_idx], schedules[2][t_idx], b, device=pos.device)
                        batch["ligand"].pos = pos
                        if crop is not None:     # sampling.py:104-109: crop at 3*tr_sigma + crop_beyond, applied on the device
                            t = float(schedules[0][t_idx])
                            model.set_crop_cutoff(cfg.tr_sigma_min ** (1 - t) * cfg.tr_sigma_max ** t * 3 + crop)
                        tr, rot, tor = model(batch)[:3]
                        # NaN guard + update formulas (sampling.py:117-186) on the device, same kernel as the native loop
                        trp, rotp, torp = model.perturb(batch, tr, rot, tor, t_idx, inference_steps, schedules, noise=z, seed=seed,
                                                        sample_ids=ids, ode=ode, no_random=no_random,
                                                        no_final_step_noise=no_final_step_noise, temp_sampling=temp_sampling,
                                                        temp_psi=temp_psi, temp_sigma_data=temp_sigma_data)
                        pos = model.modify_conformer_batch(pos, batch, trp, rotp, torp)
                finally:
                    if crop is not None:
                        model.set_crop_cutoff(None)
            pos = pos.reshape(b, n, 3)
            for i in range(b):
                data_list[lo + i]["ligand"].pos = pos[i]
            if confidence_model is not None:   # sampling.py:208-227
                if conf_batches is not None:
                    cgraphs = next(conf_batches)
                    alive = list(range(b))
                    if conf_crop is not None:   # sampling.py:213-217: every confidence graph cropped around ITS final pose
                        cgraphs = [g_.clone() for g_ in cgraphs]
                        alive = []
                        for i, g_ in enumerate(cgraphs):
                            g_["ligand"].pos = pos[i].detach().to(g_["receptor"].pos.device, g_["receptor"].pos.dtype)
                            try:
                                crop_beyond(g_, conf_crop, bool(getattr(confidence_model_args, "all_atoms", False)))
                                alive.append(i)
                            except ValueError:
                                # A pose that flew farther than the cutoff from EVERY residue has no receptor graph left to score.
                                # The reference hands such an empty graph to the confidence model; the built path cannot, so that
                                # ONE pose gets NaN (-> -1000 below, the value nan_to_num gives a failed pose) and the others of
                                # the batch are scored as usual.  Only a batch without any scorable pose raises.
                                pass
                        if not alive:
                            raise ValueError(f"crop_beyond({conf_crop}) removes every residue of every pose of the batch: nothing to score")
                    cbatch = _collate([cgraphs[i] for i in alive])
                    pos_alive = pos if len(alive) == b else pos[torch.as_tensor(alive, device=pos.device)]
                    cbatch["ligand"].pos = pos_alive.reshape(len(alive) * n, 3).to(cbatch["ligand"].pos.device)
                    if device is not None:
                        cbatch = cbatch.to(device)
                    set_time(cbatch, 0, 0, 0, len(alive), device=cbatch["ligand"].pos.device)
                    out = confidence_model(cbatch)
                    if len(alive) != b:   # graph-level outputs back in batch order, NaN for the poses that could not be scored
                        def scatter_rows(o):
                            full = torch.full((b,) + tuple(o.shape[1:]), float("nan"), device=o.device, dtype=o.dtype)
                            full[torch.as_tensor(alive, device=o.device)] = o
                            return full
                        out = (scatter_rows(out[0]),) + tuple(out[1:]) if isinstance(out, tuple) else scatter_rows(out)
                else:   # the sampling batch itself, still carrying the last step's times (sampling.py:113-114, 223)
                    batch["ligand"].pos = pos.reshape(b * n, 3)
                    set_time(batch, schedules[0][-1], schedules[1][-1], schedules[2][-1], b, device=batch["ligand"].pos.device)
                    out = confidence_model(batch)
                confidence.append(out[0] if isinstance(out, tuple) else out)
    if confidence is not None:
        confidence = torch.nan_to_num(torch.cat(confidence, dim=0), nan=-1000)
    return data_list, confidence
