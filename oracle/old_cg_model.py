"""Oracle restatement of the LEGACY score-model class: models/old_cg_model.py CGOldModel.forward :203-352, in
confidence mode (:203-291) and in score mode (read-outs :293-352, the same modules as the new classes) (with OldAtomEncoder models/layers.py:70-118 and OldTensorProductConvLayer models/tensor_layers.py:338-380) --
the class `get_model(..., old=True, confidence_mode=True)` builds, i.e. what `old_confidence_model: true` in
default_inference_args.yaml selects for the released DiffDock-L confidence checkpoint.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Differences from the new classes that the restatement keeps literally:
  * four separate convolution layers per interaction layer (ligand intra, rec->lig, receptor intra, lig->rec), each with its
    own mean over ITS edges and its own BatchNorm, summed onto the padded node features (:249-288);
  * the lig->rec layer concatenates [edge_attr, lig, rec] = [edge, GATHER node, TARGET node] (:263), the other three
    [edge, target, gather]; it reuses the forward spherical harmonics;
  * the receptor node and edge embeddings see the sigma embedding (:399-416);
  * OldAtomEncoder slices the scalar features right after the categorical ones and the language-model block from the END
    of the row -- with rows laid out [categorical | ESM | sigma] that is ESM[:sigma_dim] and [ESM[sigma_dim:] | sigma].
"""
import torch

from .cg_model import CGModelOracle
from .e3nn_lite import FullTensorProduct, Irreps
from .layers import TPConv, linear, mlp2


class CGOldOracle(CGModelOracle):
    def __init__(self, cfg, state_dict, so3_table=None, torus_table=None, dtype=torch.float32):
        assert cfg.old and cfg.sh_lmax == 2
        self.cfg, self.dtype = cfg, dtype
        self.so3_table, self.torus_table = so3_table, torus_table
        self.sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in state_dict.items()}
        self.sh_irreps = Irreps.spherical_harmonics(2)
        seq_cfg = cfg.replace(reduce_pseudoscalars=False)
        mk = lambda fam, l: TPConv(self.sd, f"{fam}.{l}", seq_cfg.layer_irreps(l)[0], self.sh_irreps, seq_cfg.layer_irreps(l)[1],
                                   residual=False, batch_norm=cfg.batch_norm, faster=False, edge_groups=1)
        L = cfg.num_conv_layers
        self.lig_conv = [mk("lig_conv_layers", l) for l in range(L)]
        self.rec_conv = [mk("rec_conv_layers", l) for l in range(L)]
        self.l2r_conv = [mk("lig_to_rec_conv_layers", l) for l in range(L)]
        self.r2l_conv = [mk("rec_to_lig_conv_layers", l) for l in range(L)]
        if not cfg.confidence_mode:   # old_cg_model.py:156-200
            last_out = seq_cfg.layer_irreps(L - 1)[1]
            self.final_conv = TPConv(self.sd, "final_conv", last_out, self.sh_irreps,
                                     "2x1o + 2x1e" if not cfg.odd_parity else "1x1o + 1x1e", residual=False, batch_norm=cfg.batch_norm)
            if not cfg.no_torsion:
                self.final_tp_tor = FullTensorProduct(self.sh_irreps, "2e")
                self.tor_bond_conv = TPConv(self.sd, "tor_bond_conv", last_out, self.final_tp_tor.irreps_out,
                                            f"{cfg.ns}x0o + {cfg.ns}x0e" if not cfg.odd_parity else f"{cfg.ns}x0o",
                                            residual=False, batch_norm=cfg.batch_norm)

    def old_atom_encoder(self, name, x, n_cat, lm_dim):
        sd, c = self.sd, self.cfg
        emb = 0
        for i in range(n_cat):
            emb = emb + sd[f"{name}.atom_embedding_list.{i}.weight"][x[:, i].long()]
        emb = emb + linear(sd, f"{name}.linear", x[:, n_cat:n_cat + c.sigma_embed_dim])
        if lm_dim:
            emb = linear(sd, f"{name}.lm_embedding_layer", torch.cat([emb, x[:, -lm_dim:]], 1))
        return emb

    def build_rec_conv_graph_old(self, data):
        """old_cg_model.py:394-416: node and edge attributes carry the sigma embedding."""
        from .layers import gaussian_smearing
        c, sd, rec = self.cfg, self.sd, data["receptor"]
        rec.node_sigma_emb = self._temb(rec.node_t["tr"])
        node_attr = torch.cat([rec.x.to(self.dtype), rec.node_sigma_emb], 1)
        pos = rec.pos.to(self.dtype)
        ei = data["receptor", "receptor"].edge_index
        vec = pos[ei[1]] - pos[ei[0]]
        edge_attr = torch.cat([rec.node_sigma_emb[ei[0]], gaussian_smearing(sd["rec_distance_expansion.offset"], vec.norm(dim=-1))], 1)
        return node_attr, ei, edge_attr, self._sh(vec), self._edge_weight(vec, c.rec_max_radius)

    def __call__(self, data, return_intermediates=False):
        c, sd, ns = self.cfg, self.sd, self.cfg.ns
        if c.no_aminoacid_identities:   # old_cg_model.py:204-205
            data["receptor"].x = data["receptor"].x * 0
        tr_sigma, rot_sigma, tor_sigma = self._sigmas(data)     # raw t in confidence mode, t_to_sigma(t) in score mode (:207-210)
        lig_node_attr, lig_ei, lig_edge_attr, lig_sh, lig_ew = self.build_lig_conv_graph(data)
        lig_node_attr = self.old_atom_encoder("lig_node_embedding", lig_node_attr, 16, 0)
        lig_edge_attr = mlp2(sd, "lig_edge_embedding", lig_edge_attr)
        rec_node_attr, rec_ei, rec_edge_attr, rec_sh, rec_ew = self.build_rec_conv_graph_old(data)
        rec_node_attr = self.old_atom_encoder("rec_node_embedding", rec_node_attr, 1, c.lm_embedding_dim)
        rec_edge_attr = mlp2(sd, "rec_edge_embedding", rec_edge_attr)
        cutoff = (tr_sigma * 3 + 20).unsqueeze(1).to(self.dtype) if c.dynamic_max_cross else c.cross_max_distance
        lr_ei, lr_edge_attr, lr_sh, _, lr_ew = self.build_cross_conv_graph(data, cutoff)
        lr_edge_attr = mlp2(sd, "cross_edge_embedding", lr_edge_attr)
        cross_lig, cross_rec = lr_ei
        inter = {"edge_counts": (lig_ei.shape[1], lr_ei.shape[1], rec_ei.shape[1])} if return_intermediates else None
        L = c.num_conv_layers
        pad = lambda x, w: torch.nn.functional.pad(x, (0, w - x.shape[-1]))
        for l in range(L):
            ea = torch.cat([lig_edge_attr, lig_node_attr[lig_ei[0], :ns], lig_node_attr[lig_ei[1], :ns]], -1)
            lig_intra = self.lig_conv[l](lig_node_attr, lig_ei, ea, lig_sh, edge_weight=lig_ew)
            ea = torch.cat([lr_edge_attr, lig_node_attr[cross_lig, :ns], rec_node_attr[cross_rec, :ns]], -1)
            lig_inter = self.r2l_conv[l](rec_node_attr, lr_ei, ea, lr_sh, out_nodes=lig_node_attr.shape[0], edge_weight=lr_ew)
            if l != L - 1:
                ea = torch.cat([rec_edge_attr, rec_node_attr[rec_ei[0], :ns], rec_node_attr[rec_ei[1], :ns]], -1)
                rec_intra = self.rec_conv[l](rec_node_attr, rec_ei, ea, rec_sh, edge_weight=rec_ew)
                ea = torch.cat([lr_edge_attr, lig_node_attr[cross_lig, :ns], rec_node_attr[cross_rec, :ns]], -1)
                rl = self.l2r_conv[l](lig_node_attr, torch.flip(lr_ei, dims=[0]), ea, lr_sh, out_nodes=rec_node_attr.shape[0],
                                      edge_weight=lr_ew)
            lig_node_attr = pad(lig_node_attr, lig_intra.shape[-1]) + lig_intra + lig_inter
            if l != L - 1:
                rec_node_attr = pad(rec_node_attr, rec_intra.shape[-1]) + rec_intra + rl
            if inter is not None:
                inter[f"lig{l + 1}"], inter[f"rec{l + 1}"] = lig_node_attr.clone(), rec_node_attr.clone()
        if not c.confidence_mode:   # the legacy class returns a 3-tuple (:329,352)
            out = self._readouts(data, lig_node_attr, tr_sigma, rot_sigma, tor_sigma, inter, return_intermediates)
            return out[:3] + (out[4:] if return_intermediates else ())
        x = torch.cat([lig_node_attr[:, :ns], lig_node_attr[:, -ns:]], 1) if L >= 3 else lig_node_attr[:, :ns]
        batch = data["ligand"].batch
        x = torch.zeros(data.num_graphs, x.shape[1], dtype=x.dtype).index_add_(0, batch, x) / \
            torch.bincount(batch, minlength=data.num_graphs).clamp(min=1).unsqueeze(1).to(x.dtype)

        def bn1d(i, v):
            p = f"confidence_predictor.{i}"
            return (v - sd[p + ".running_mean"]) / torch.sqrt(sd[p + ".running_var"] + 1e-5) * sd[p + ".weight"] + sd[p + ".bias"]
        lin = lambda i, v: torch.nn.functional.linear(v, sd[f"confidence_predictor.{i}.weight"], sd[f"confidence_predictor.{i}.bias"])
        x = torch.relu(bn1d(1, lin(0, x)))
        x = torch.relu(bn1d(5, lin(4, x)))
        out = lin(8, x).squeeze(dim=-1)
        return (out, inter) if return_intermediates else out


CGOldConfidenceOracle = CGOldOracle   # earlier name (confidence mode only)
