"""`MIScoreModel`: drop-in for the reference score model object.

The reference boundary for this path is a Python duck type:
    model = get_model(args, device, t_to_sigma, no_parallel=True)      utils/utils.py:172
    model.load_state_dict(sd, strict=True); model.to(device); model.eval()   inference.py:202-205
    tr, rot, tor = model(batch)[:3]                                    utils/sampling.py:116
`MIScoreModel` keeps that surface (same state_dict keys, same call signature, same return
tuple `(tr[B,3], rot[B,3], tor[sum R], None)`) and forwards to libddmi.so through ctypes.
PyTorch is used only as the owner of device memory and of the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import lib as _lib
from .config import ModelConfig, config_from_args
from .hetero import as_numpy_mask


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def time_frequencies(sigma_embed_dim: int) -> torch.Tensor:
    """Frequencies of sinusoidal_embedding, computed with the reference's exact fp32 op sequence
    (utils/diffusion_utils.py:101-103) so the device phases are bit-identical."""
    half = sigma_embed_dim // 2
    emb = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32) * -emb)


def _mask_fingerprint(mr):
    """Key of the rotatable-bond masks (numpy arrays / tensors, possibly a list per graph).  Host arrays are keyed by CONTENT (an
    id() can be recycled after the old list is collected, and in-place edits leave it unchanged; hashing them is a host-side
    memcpy).  DEVICE tensors are keyed by (storage address, in-place version counter, shape): hashing their content would put a
    blocking device-to-host copy in front of every forward of the step-wise loop."""
    import hashlib
    import numpy as np
    if mr is None:
        return None
    h = hashlib.sha1()
    for a in (mr if isinstance(mr, (list, tuple)) else [mr]):
        if torch.is_tensor(a) and a.device.type != "cpu":
            h.update(repr((a.data_ptr(), 0 if a.is_inference() else a._version, tuple(a.shape), str(a.dtype))).encode())
            continue
        a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
        h.update(str(a.shape).encode())
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes() if a.size else b"")
    return h.hexdigest()


class MIScoreModel:
    def __init__(self, cfg: ModelConfig, device="cuda:0", lib_path: str | None = None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = _lib.load(lib_path)
        self._h = C.c_void_p()
        dev_index = self.device.index or 0 if self.device.type == "cuda" else 0
        _lib.check(self.lib, self.lib.ddmi_create(C.byref(_lib.make_config(cfg)), dev_index, C.byref(self._h)))
        self._complex_key = None
        self._complex_ref = None
        self._keep = None
        self._state: Dict[str, torch.Tensor] = {}
        self._tables_set = False
        self.training = False

    # ------------------------------------------------------------------ nn.Module-like surface
    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device).type == self.device.type, "a handle is bound to its device at construction"
        return self

    def parameters(self):
        return iter(self._state.values())

    def state_dict(self):
        return dict(self._state)

    def expected_keys(self):
        n = self.lib.ddmi_num_weights(self._h)
        out = {}
        for i in range(n):
            key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            _lib.check(self.lib, self.lib.ddmi_weight_spec(self._h, i, C.byref(key), shape, C.byref(nd)))
            out[key.value.decode()] = tuple(shape[:nd.value])
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        expected = self.expected_keys()
        # e3nn keeps constant buffers under *.tp.* / final_tp_tor.* in real checkpoints, and nn.BatchNorm1d (confidence
        # predictors, cg_model.py:184-207) its `num_batches_tracked` step counter: neither is a weight
        # ... e3nn's o3.Linear registers an `output_mask` buffer (and an empty `bias`) next to `weight`; a confidence-mode module
        # built with sidechain_pred owns a sidechain_predictor its forward never calls (cg_model.py:173-178 vs 353-366)
        given = {k: v for k, v in sd.items() if ".tp." not in k and not k.startswith("final_tp_tor.")
                 and not k.endswith("num_batches_tracked") and not k.endswith(".output_mask")
                 and not (k.endswith((".linear_2.bias", "sidechain_predictor.bias")) and v.numel() == 0)
                 and not (self.cfg.confidence_mode and k.startswith("sidechain_predictor."))}
        missing = [k for k in expected if k not in given]
        unexpected = [k for k in given if k not in expected]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        freq = time_frequencies(self.cfg.sigma_embed_dim).contiguous()
        _lib.check(self.lib, self.lib.ddmi_set_time_frequencies(self._h, _ptr(freq), freq.numel()))
        for k in expected:
            if k not in given:
                continue
            t = given[k].detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check(self.lib, self.lib.ddmi_set_weight(self._h, k.encode(), _ptr(t), shape, t.dim()))
            self._state[k] = t
        _lib.check(self.lib, self.lib.ddmi_commit_weights(self._h))
        self.invalidate_complex()
        return self

    def invalidate_complex(self):
        """Forget the cached ddmi_set_complex: the next model(batch) rebuilds the static part of the batch."""
        self._complex_key = None
        self._complex_ref = None

    def set_tables(self, so3_exp_score_norms: np.ndarray, torus_score_norm: np.ndarray):
        """Score-norm tables (utils/so3.py:59, utils/torus.py:72-76); see diffdock_amd/tables.py."""
        for kind, tab in ((0, so3_exp_score_norms), (1, torus_score_norm)):
            tab = np.ascontiguousarray(tab, dtype=np.float64)
            _lib.check(self.lib, self.lib.ddmi_set_table(self._h, kind, tab.ctypes.data, tab.size))
        self._tables_set = True
        return self

    def __del__(self):
        try:
            if self._h:
                self.lib.ddmi_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ batch -> ddmi_complex
    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _static_tensors(self, data):
        lig, rec = data["ligand"], data["receptor"]
        bond, rr = data["ligand", "ligand"], data["receptor", "receptor"]
        ts = [rec.x, rec.pos, lig.x, lig.batch, rec.batch, lig.edge_mask, bond.edge_index, bond.edge_attr, rr.edge_index]
        if self.cfg.all_atoms:
            ts += [data["atom"].x, data["atom"].pos, data["atom"].batch, data["atom", "atom"].edge_index, data["atom", "receptor"].edge_index]
        return ts

    def _ensure_complex(self, data):
        """ddmi_set_complex once per batch OBJECT.  One model is reused over many complexes (inference.py:224-303) and the
        caching allocator hands a freed batch's addresses to the next one, so addresses and shapes cannot identify a
        batch: the cache is keyed on the identity of the live batch object (weak reference) plus the in-place version
        counters of its static tensors; a batch type that cannot be weakly referenced is fingerprinted by content."""
        lig, rec = data["ligand"], data["receptor"]
        bond, rr = data["ligand", "ligand"], data["receptor", "receptor"]
        static = self._static_tensors(data)
        # (tensors created under torch.inference_mode() carry no version counter: their slot of the key is 0)
        key = tuple((t.data_ptr(), 0 if t.is_inference() else t._version, tuple(t.shape)) for t in static) + (int(data.num_graphs),)
        # rotatable-bond masks (numpy arrays / tensors, possibly a list per graph): keyed by CONTENT -- an id() can be recycled
        # after the old list is collected, and in-place edits leave it unchanged
        key = key + (_mask_fingerprint(getattr(lig, "mask_rotate", None)),)
        same_obj = self._complex_ref is not None and self._complex_ref() is data
        if same_obj and key == self._complex_key:
            return
        try:
            ref = weakref.ref(data)
        except TypeError:    # no weak references: content fingerprint of the static tensors (one host read-back per call)
            ref = None
            key = key + tuple(float(t.double().sum()) + float((t.double() * t.double()).sum()) for t in static)
            if key == self._complex_key:
                return
        if not self._tables_set:
            from .tables import default_tables
            self.set_tables(*default_tables())
        dev = self.device
        B = int(data.num_graphs)

        def ptr_of(batch_vec, n):
            cnt = torch.bincount(batch_vec.to("cpu"), minlength=B)
            p = torch.zeros(B + 1, dtype=torch.int32)
            p[1:] = torch.cumsum(cnt, 0)
            assert int(p[-1]) == n
            return p
        lig_ptr, rec_ptr = ptr_of(lig.batch, lig.pos.shape[0]), ptr_of(rec.batch, rec.pos.shape[0])
        i32 = lambda t: t.to(dev, torch.int32).contiguous()
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        edge_mask = lig.edge_mask.to(dev, torch.uint8).contiguous()
        n_tor = int(lig.edge_mask.sum())
        mask_rotate = None
        mr_attr = getattr(lig, "mask_rotate", None) if hasattr(lig, "mask_rotate") else None
        if mr_attr is not None:
            mr = as_numpy_mask(mr_attr)
            if mr.size and mr.shape[0] * B == n_tor and mr.shape[1] * B == lig.pos.shape[0]:
                mask_rotate = torch.from_numpy(np.ascontiguousarray(mr.astype(np.uint8))).to(dev)
        keep = dict(lig_ptr=lig_ptr, rec_ptr=rec_ptr, lig_x=i32(lig.x[:, :16]), bond_index=i32(bond.edge_index),
                    bond_attr=f32(bond.edge_attr), edge_mask=edge_mask,
                    rec_x=f32(rec.x * 0 if self.cfg.no_aminoacid_identities else rec.x), rec_pos=f32(rec.pos),   # cg_model.py:309-310
                    rec_edge_index=i32(rr.edge_index), mask_rotate=mask_rotate)
        c = _lib.Complex()
        c.num_graphs, c.n_lig, c.n_rec = B, lig.pos.shape[0], rec.pos.shape[0]
        c.n_bond_edges, c.n_rec_edges, c.n_tor = bond.edge_index.shape[1], rr.edge_index.shape[1], n_tor
        names = ["lig_ptr", "rec_ptr", "lig_x", "bond_index", "bond_attr", "edge_mask", "rec_x", "rec_pos",
                 "rec_edge_index", "mask_rotate"]
        if self.cfg.all_atoms:   # receptor heavy atoms and their static relations (models/aa_model.py:291-303)
            atom, aa, ar = data["atom"], data["atom", "atom"], data["atom", "receptor"]
            keep.update(atom_ptr=ptr_of(atom.batch, atom.pos.shape[0]), atom_x=i32(atom.x[:, :4]), atom_pos=f32(atom.pos),
                        atom_edge_index=i32(aa.edge_index), atom_rec_edge_index=i32(ar.edge_index))
            c.n_atom, c.n_atom_edges, c.n_atom_rec_edges = atom.pos.shape[0], aa.edge_index.shape[1], ar.edge_index.shape[1]
            names += ["atom_ptr", "atom_x", "atom_pos", "atom_edge_index", "atom_rec_edge_index"]
        for name in names:
            setattr(c, name, keep[name].data_ptr() if keep[name] is not None else None)
        _lib.check(self.lib, self.lib.ddmi_set_complex(self._h, C.byref(c), self._stream()))
        self._keep, self._complex_key, self._complex_ref = keep, key, ref
        self._B, self._n_tor, self._n_lig = B, n_tor, lig.pos.shape[0]

    # ------------------------------------------------------------------ model(batch)
    def __call__(self, data):
        self._ensure_complex(data)
        dev = self.device
        pos = data["ligand"].pos.to(dev, torch.float32).contiguous()
        t = [data.complex_t[k].to(dev, torch.float32).contiguous() for k in ("tr", "rot", "tor")]
        if self.cfg.confidence_mode:   # (confidence, atom_confidence) -- models/cg_model.py:353-366
            n_out = self.cfg.num_confidence_outputs + (1 if self.cfg.affinity_prediction else 0)
            conf = torch.empty(self._B, n_out, device=dev)
            atom = torch.empty(self._n_lig, self.cfg.atom_num_confidence_outputs, device=dev) if self.cfg.atom_confidence else None
            _lib.check(self.lib, self.lib.ddmi_confidence(self._h, _ptr(pos), _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(conf),
                                                          None if atom is None else _ptr(atom), self._stream()))
            if self.cfg.old:   # the legacy class returns the bare tensor (old_cg_model.py:287-291)
                return conf.squeeze(-1)
            return conf.squeeze(-1), (atom if atom is not None else torch.zeros(self._n_lig, device=dev))
        tr = torch.empty(self._B, 3, device=dev)
        rot = torch.empty(self._B, 3, device=dev)
        no_tor = self.cfg.no_torsion or self._n_tor == 0
        tor = torch.empty(0 if no_tor else self._n_tor, device=dev)
        _lib.check(self.lib, self.lib.ddmi_forward(self._h, _ptr(pos), _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(tr),
                                                   _ptr(rot), None if no_tor else _ptr(tor), self._stream()))
        if self.cfg.old:   # the legacy class returns a 3-tuple (old_cg_model.py:329,352)
            return tr, rot, tor
        side = None
        if self.cfg.sidechain_pred:   # models/cg_model.py:397-402: [n_rec, 10]
            side = torch.empty(int(data["receptor"].pos.shape[0]), 10, device=dev)
            _lib.check(self.lib, self.lib.ddmi_sidechain_pred(self._h, _ptr(side), self._stream()))
            if getattr(self, "_crop_cutoff", 0.0) > 0.0:   # the reference crops the graph first: rows of the kept residues only
                side = side[torch.from_numpy(self.debug_buffer("crop_keep") != 0).to(dev)]
        return tr, rot, tor, side

    forward = __call__

    def set_crop_cutoff(self, cutoff):
        """crop_beyond(graph, cutoff) for the following model(batch) calls (utils/utils.py:388-413); None / 0 = off."""
        self._crop_cutoff = float(cutoff or 0.0)
        _lib.check(self.lib, self.lib.ddmi_set_crop_cutoff(self._h, self._crop_cutoff))

    def modify_conformer_batch(self, pos, data, tr_update, rot_update, torsion_updates, mask_rotate=None):
        """utils/diffusion_utils.py:60-78 on the device (same argument order; mask_rotate comes from the batch)."""
        self._ensure_complex(data)
        dev = self.device
        out = pos.to(dev, torch.float32).contiguous().clone()
        f = lambda x: None if x is None else x.to(dev, torch.float32).contiguous()
        tr_u, rot_u, tor_u = f(tr_update), f(rot_update), f(torsion_updates)
        _lib.check(self.lib, self.lib.ddmi_modify_conformer(self._h, _ptr(out), _ptr(tr_u), _ptr(rot_u), _ptr(tor_u),
                                                            self._stream()))
        return out

    def _sample_cfg(self, inference_steps, schedules, noise, seed, sample_ids, ode, no_random, no_final_step_noise,
                    temp_sampling, temp_psi, temp_sigma_data, crop_beyond):
        """ddmi_sample_cfg + the host / device arrays it points into (returned so that they outlive the call)."""
        dev = self.device
        sc = _lib.SampleCfg()
        sched = [np.ascontiguousarray(np.asarray(s, dtype=np.float64)) for s in schedules]
        sc.inference_steps = inference_steps
        sc.tr_schedule, sc.rot_schedule, sc.tor_schedule = (s.ctypes.data for s in sched)
        sc.ode, sc.no_random, sc.no_final_step_noise = int(ode), int(no_random), int(no_final_step_noise)
        three = lambda v: list(v) if hasattr(v, "__iter__") else [v] * 3
        for name, v in (("temp_sampling", temp_sampling), ("temp_psi", temp_psi), ("temp_sigma_data", temp_sigma_data)):
            arr = getattr(sc, name)
            for i, x in enumerate(three(v)):
                arr[i] = float(x)
        sc.seed = int(seed)
        sc.use_crop, sc.crop_beyond = (0, 0.0) if crop_beyond is None else (1, float(crop_beyond))
        ids = None
        if sample_ids is not None:
            ids = np.ascontiguousarray(np.asarray(sample_ids, dtype=np.int64))
            assert ids.size == self._B, "one sample id per graph of the batch"
            sc.sample_ids = ids.ctypes.data
        zs = [None, None, None]
        if noise is not None:
            zs = [None if z is None else z.to(dev, torch.float32).contiguous() for z in noise]
            sc.z_tr, sc.z_rot, sc.z_tor = (None if z is None else z.data_ptr() for z in zs)
        return sc, (sched, ids, zs)

    def sample_batch(self, data, inference_steps, schedules, noise=None, seed=0, sample_ids=None, ode=False,
                     no_random=False, no_final_step_noise=False, temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5,
                     crop_beyond=None):
        """The whole step loop of sampling() (utils/sampling.py:96-191) for one collated batch, on the device."""
        self._ensure_complex(data)
        pos = data["ligand"].pos.to(self.device, torch.float32).contiguous().clone()
        sc, keep = self._sample_cfg(inference_steps, schedules, noise, seed, sample_ids, ode, no_random, no_final_step_noise,
                                    temp_sampling, temp_psi, temp_sigma_data, crop_beyond)
        _lib.check(self.lib, self.lib.ddmi_sample(self._h, _ptr(pos), C.byref(sc), self._stream()))
        if noise is not None and self.device.type == "cuda":   # the injected draws must outlive the enqueued steps
            for z in keep[2]:
                if z is not None:
                    z.record_stream(torch.cuda.current_stream(self.device))
        return pos

    def perturb(self, data, tr_score, rot_score, tor_score, t_idx, inference_steps, schedules, noise=None, seed=0,
                sample_ids=None, ode=False, no_random=False, no_final_step_noise=False, temp_sampling=1.0, temp_psi=0.0,
                temp_sigma_data=0.5):
        """Scores of step t_idx -> (tr_perturb, rot_perturb, tor_perturb): NaN guard + update formulas of
        utils/sampling.py:117-186 on the device (ddmi_perturb).  `noise` = (z_tr [steps,B,3], z_rot [steps,B,3], z_tor [steps,n_tor])."""
        self._ensure_complex(data)
        dev = self.device
        f = lambda x: x.to(dev, torch.float32).contiguous().clone()
        tr, rot = f(tr_score), f(rot_score)
        tor = f(tor_score) if (tor_score is not None and tor_score.numel()) else None
        sc, keep = self._sample_cfg(inference_steps, schedules, noise, seed, sample_ids, ode, no_random, no_final_step_noise,
                                    temp_sampling, temp_psi, temp_sigma_data, None)
        _lib.check(self.lib, self.lib.ddmi_perturb(self._h, _ptr(tr), _ptr(rot), _ptr(tor), C.byref(sc), int(t_idx), self._stream()))
        if noise is not None and self.device.type == "cuda":
            for z in keep[2]:
                if z is not None:
                    z.record_stream(torch.cuda.current_stream(self.device))
        return tr, rot, tor

    # ------------------------------------------------------------------ introspection
    def set_kernel_timing(self, enabled, level: int | None = None):
        """level 1 = one row per kernel name, 2 = k_conv_fused per edge group, 3 = per (layer, edge group); the harness variable
        DDMI_TIME_GROUPS=1|2 selects level 2 | 3 (profiling scripts under tools/)."""
        if level is None:
            tg = os.environ.get("DDMI_TIME_GROUPS")
            level = 1 if not tg else (3 if int(tg) >= 2 else 2)
        _lib.check(self.lib, self.lib.ddmi_set_kernel_timing(self._h, level if enabled else 0))

    def kernel_timings(self):
        """{phase: (total_ms, launches)} measured with HIP events on the launch stream."""
        out, i = {}, 0
        while True:
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
            if self.lib.ddmi_kernel_timings(self._h, i, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                break
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    def debug_buffer(self, name: str) -> np.ndarray:
        shape, nd, is_int = (C.c_int64 * 4)(), C.c_int(), C.c_int()
        _lib.check(self.lib, self.lib.ddmi_debug_shape(self._h, name.encode(), shape, C.byref(nd), C.byref(is_int)))
        arr = np.empty(tuple(shape[:nd.value]), dtype=np.int32 if is_int.value else np.float32)
        _lib.check(self.lib, self.lib.ddmi_debug_read(self._h, name.encode(), arr.ctypes.data, arr.nbytes, self._stream()))
        return arr


def get_model(args, device, t_to_sigma=None, no_parallel=True, confidence_mode=False, old=False, lib_path=None):
    """Same signature as the reference factory (utils/utils.py:172).  `t_to_sigma` is accepted for
    compatibility; the geometric schedule it implements (utils/diffusion_utils.py:28-32) is evaluated
    in-library from the sigma bounds in `args`."""
    cfg = args if isinstance(args, ModelConfig) else config_from_args(args)
    if old:   # utils/utils.py:180-219: CGOldModel; no sh_lmax / embedding-layer / pseudoscalar arguments reach it
        if cfg.all_atoms:
            raise NotImplementedError("the legacy classes are built on CG graphs only (models/old_cg_model.py)")
        cfg = cfg.replace(old=True, confidence_mode=bool(confidence_mode), sh_lmax=2, num_prot_emb_layers=0,
                          depthwise_convolution=False, sidechain_pred=False,   # (not among the arguments get_model(old=True) passes, utils/utils.py:180-219)
                          reduce_pseudoscalars=False, num_confidence_outputs=1,
                          use_old_atom_encoder=getattr(args, "use_old_atom_encoder", True) if not isinstance(args, ModelConfig)
                          else cfg.use_old_atom_encoder)
        if not cfg.use_old_atom_encoder:
            raise NotImplementedError("CGOldModel with the new AtomEncoder cannot be constructed by the reference either "
                                      "(AtomEncoder has no lm_embedding_type argument)")
    elif not cfg.embed_also_ligand and not cfg.all_atoms:
        # CGModel.ligand_embedding asserts it on every forward (models/cg_model.py:263, "otherwise reimplement padding").
        # AAModel has no such assert: without ligand embedding layers it zero-pads the ligand rows to the receptor width
        # (models/aa_model.py:351-357) -- a no-op when num_prot_emb_layers == 0, which is the case accepted here (config.py)
        raise AssertionError("embed_also_ligand must be set for the CGModel class (models/cg_model.py:263)")
    if confidence_mode != cfg.confidence_mode:
        cfg = cfg.replace(confidence_mode=bool(confidence_mode))
    return MIScoreModel(cfg, device=device, lib_path=lib_path)
