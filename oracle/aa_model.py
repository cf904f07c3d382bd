"""Oracle restatement of models/aa_model.py AAModel.forward (score mode, eval): the all-atom variant of the score
model -- receptor heavy atoms as a third node type, nine edge groups per interaction layer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Line map (reference models/aa_model.py):
  forward :364-484 (node table [lig; rec; atom], groups ll, lr, la, rr, rl, ra, aa, al, ar :399-427; the flipped groups
  REUSE the forward spherical harmonics :411-412, unlike CGModel)      embedding :275-362
  build_atom_conv_graph :568-582   build_cross_lig_conv_graph :584-621   build_cross_rec_conv_graph :623-633
Ligand graph, centre/bond graphs and the read-outs are the CGModel ones (same code in the reference).
"""
import torch
import torch.nn.functional as F

from .cg_model import CGModelOracle
from .conformer import t_to_sigma
from .graph_ops import radius
from .layers import TPConv, atom_encoder, gaussian_smearing, mlp2


class AAModelOracle(CGModelOracle):
    def __init__(self, cfg, state_dict, so3_table, torus_table, dtype=torch.float32):
        super().__init__(cfg, state_dict, so3_table, torus_table, dtype)
        c, sd = cfg, self.sd
        g4 = 4 if c.differentiate_convolutions else 1
        self.rec_emb_layers = [TPConv(sd, f"rec_emb_layers.{i}", *self._io(i), residual=True, batch_norm=c.batch_norm,
                                      faster=c.faster, edge_groups=g4, tp_weights_layers=c.tp_weights_layers)
                               for i in range(c.num_prot_emb_layers)]

    # ------------------------------------------------------------------ receptor-side graphs (static)
    def build_atom_conv_graph(self, data):
        c, sd, atom = self.cfg, self.sd, data["atom"]
        pos = atom.pos.to(self.dtype)
        src, dst = data["atom", "atom"].edge_index
        vec = pos[dst] - pos[src]
        edge_attr = gaussian_smearing(sd["lig_distance_expansion.offset"], vec.norm(dim=-1))
        return atom.x.to(self.dtype), edge_attr, self._sh(vec), self._edge_weight(vec, c.lig_max_radius)

    def build_cross_rec_conv_graph(self, data):
        sd = self.sd
        ar = data["atom", "receptor"].edge_index
        vec = data["receptor"].pos.to(self.dtype)[ar[1]] - data["atom"].pos.to(self.dtype)[ar[0]]
        return gaussian_smearing(sd["rec_distance_expansion.offset"], vec.norm(dim=-1)), self._sh(vec), 1

    def build_cross_lig_conv_graph(self, data, cutoff):
        c, sd, lig, rec, atom = self.cfg, self.sd, data["ligand"], data["receptor"], data["atom"]
        lpos, rpos, apos = lig.pos.to(self.dtype), rec.pos.to(self.dtype), atom.pos.to(self.dtype)
        if torch.is_tensor(cutoff):
            lr = radius(rpos / cutoff[rec.batch], lpos / cutoff[lig.batch], 1, rec.batch, lig.batch, max_num_neighbors=10000)
        else:
            lr = radius(rpos, lpos, cutoff, rec.batch, lig.batch, max_num_neighbors=10000)
        vec = rpos[lr[1]] - lpos[lr[0]]
        lr_attr = torch.cat([lig.node_sigma_emb[lr[0]],
                             gaussian_smearing(sd["cross_distance_expansion.offset"], vec.norm(dim=-1))], 1)
        cutoff_d = cutoff[lig.batch[lr[0]]].squeeze() if torch.is_tensor(cutoff) else cutoff
        lr_sh, lr_ew = self._sh(vec), self._edge_weight(vec, cutoff_d)
        la = radius(apos, lpos, c.lig_max_radius, atom.batch, lig.batch, max_num_neighbors=10000)
        vec = apos[la[1]] - lpos[la[0]]
        la_attr = torch.cat([lig.node_sigma_emb[la[0]],
                             gaussian_smearing(sd["lig_distance_expansion.offset"], vec.norm(dim=-1))], 1)
        return lr, lr_attr, lr_sh, lr_ew, la, la_attr, self._sh(vec), self._edge_weight(vec, c.lig_max_radius)

    # ------------------------------------------------------------------ embedding (aa_model.py:275-362)
    def embedding(self, data):
        c, ns, sd = self.cfg, self.cfg.ns, self.sd
        rec, atom = data["receptor"], data["atom"]
        rec_node_attr, rec_edge_attr, rec_edge_sh, rec_ew = self.build_rec_conv_graph(data)
        rec_node_attr = atom_encoder(sd, "rec_node_embedding", rec_node_attr, 1)
        rec_edge_attr = mlp2(sd, "rec_edge_embedding", rec_edge_attr)
        atom_node_attr, atom_edge_attr, atom_edge_sh, atom_ew = self.build_atom_conv_graph(data)
        atom_node_attr = atom_encoder(sd, "atom_node_embedding", atom_node_attr, 4)
        atom_edge_attr = mlp2(sd, "atom_edge_embedding", atom_edge_attr)
        ar_edge_attr, ar_edge_sh, ar_ew = self.build_cross_rec_conv_graph(data)
        ar_edge_attr = mlp2(sd, "ar_edge_embedding", ar_edge_attr)
        rec_ei = data["receptor", "receptor"].edge_index.clone()
        atom_ei = data["atom", "atom"].edge_index.clone()
        ar_ei = data["atom", "receptor"].edge_index.clone()
        n_rec = len(rec_node_attr)
        if self.rec_emb_layers:
            node_attr = torch.cat([rec_node_attr, atom_node_attr], 0)
            ar2 = ar_ei.clone()
            ar2[0] = ar2[0] + n_rec
            edge_index = torch.cat([rec_ei, ar2, atom_ei + n_rec, torch.flip(ar2, dims=[0])], 1)
            edge_attr = torch.cat([rec_edge_attr, ar_edge_attr, atom_edge_attr, ar_edge_attr], 0)
            edge_sh = torch.cat([rec_edge_sh, ar_edge_sh, atom_edge_sh, ar_edge_sh], 0)
            if torch.is_tensor(rec_ew):
                edge_weight = torch.cat([rec_ew, torch.ones(ar_ei.shape[1], 1, dtype=self.dtype), atom_ew,
                                         torch.ones(ar_ei.shape[1], 1, dtype=self.dtype)], 0)
            else:
                edge_weight = torch.ones(edge_index.shape[1], 1, dtype=self.dtype)
            s1 = rec_ei.shape[1]
            s2 = s1 + ar_ei.shape[1]
            s3 = s2 + atom_ei.shape[1]
            for layer in self.rec_emb_layers:
                ea = torch.cat([edge_attr, node_attr[edge_index[0], :ns], node_attr[edge_index[1], :ns]], -1)
                if c.differentiate_convolutions:
                    ea = [ea[:s1], ea[s1:s2], ea[s2:s3], ea[s3:]]
                node_attr = layer(node_attr, edge_index, ea, edge_sh, edge_weight=edge_weight)
            rec_node_attr, atom_node_attr = node_attr[:n_rec], node_attr[n_rec:]
        sig = mlp2(sd, "rec_sigma_embedding", self._temb(data.complex_t["tr"]))
        rec_node_attr = rec_node_attr.clone()
        rec_node_attr[:, :ns] = rec_node_attr[:, :ns] + sig[rec.batch]
        rec_edge_attr = rec_edge_attr + sig[rec.batch[rec_ei[0]]]
        atom_node_attr = atom_node_attr.clone()
        atom_node_attr[:, :ns] = atom_node_attr[:, :ns] + sig[atom.batch]
        atom_edge_attr = atom_edge_attr + sig[atom.batch[atom_ei[0]]]
        ar_edge_attr = ar_edge_attr + sig[atom.batch[ar_ei[0]]]
        lig = self.ligand_embedding(data)
        if not c.embed_also_ligand:   # aa_model.py:355-357: ligand rows zero-padded to the width the embedding layers produced
            lig = (F.pad(lig[0], (0, rec_node_attr.shape[-1] - lig[0].shape[-1])),) + lig[1:]
        return lig + (rec_node_attr, rec_ei, rec_edge_attr, rec_edge_sh, rec_ew,
                                               atom_node_attr, atom_ei, atom_edge_attr, atom_edge_sh, atom_ew,
                                               ar_ei, ar_edge_attr, ar_edge_sh, ar_ew)

    # ------------------------------------------------------------------ forward (aa_model.py:364-436)
    def __call__(self, data, return_intermediates=False):
        c, sd, ns = self.cfg, self.sd, self.cfg.ns
        if c.no_aminoacid_identities:   # cg_model.py:309-310 / aa_model.py:377-378
            data["receptor"].x = data["receptor"].x * 0
        tr_sigma, rot_sigma, tor_sigma = self._sigmas(data)
        (lig_node_attr, lig_ei, lig_edge_attr, lig_edge_sh, lig_ew,
         rec_node_attr, rec_ei, rec_edge_attr, rec_edge_sh, rec_ew,
         atom_node_attr, atom_ei, atom_edge_attr, atom_edge_sh, atom_ew,
         ar_ei, ar_edge_attr, ar_edge_sh, ar_ew) = self.embedding(data)
        cutoff = (tr_sigma * 3 + 20).unsqueeze(1).to(self.dtype) if c.dynamic_max_cross else c.cross_max_distance
        lr_ei, lr_edge_attr, lr_edge_sh, lr_ew, la_ei, la_edge_attr, la_edge_sh, la_ew = \
            self.build_cross_lig_conv_graph(data, cutoff)
        lr_edge_attr = mlp2(sd, "lr_edge_embedding", lr_edge_attr)
        la_edge_attr = mlp2(sd, "la_edge_embedding", la_edge_attr)
        n_lig, n_rec = len(lig_node_attr), len(rec_node_attr)
        node_attr = torch.cat([lig_node_attr, rec_node_attr, atom_node_attr], 0)
        rec_ei, atom_ei, lr_ei, la_ei, ar_ei = rec_ei + n_lig, atom_ei + n_lig + n_rec, lr_ei.clone(), la_ei.clone(), ar_ei.clone()
        lr_ei[1] += n_lig
        la_ei[1] += n_lig + n_rec
        ar_ei[0] += n_lig + n_rec
        ar_ei[1] += n_lig
        groups = [lig_ei, lr_ei, la_ei, rec_ei, torch.flip(lr_ei, dims=[0]), torch.flip(ar_ei, dims=[0]), atom_ei,
                  torch.flip(la_ei, dims=[0]), ar_ei]
        attrs = [lig_edge_attr, lr_edge_attr, la_edge_attr, rec_edge_attr, lr_edge_attr, ar_edge_attr, atom_edge_attr,
                 la_edge_attr, ar_edge_attr]
        shs = [lig_edge_sh, lr_edge_sh, la_edge_sh, rec_edge_sh, lr_edge_sh, ar_edge_sh, atom_edge_sh, la_edge_sh, ar_edge_sh]
        edge_index, edge_attr, edge_sh = torch.cat(groups, 1).long(), torch.cat(attrs, 0), torch.cat(shs, 0)
        if torch.is_tensor(lig_ew):
            one = torch.ones(ar_ei.shape[1], 1, dtype=self.dtype)
            edge_weight = torch.cat([lig_ew, lr_ew, la_ew, rec_ew, lr_ew, one, atom_ew, la_ew, one], 0)
        else:
            edge_weight = torch.ones(edge_index.shape[1], 1, dtype=self.dtype)
        cuts = [0]
        for g in groups:
            cuts.append(cuts[-1] + g.shape[1])
        inter = {"edge_counts": tuple(g.shape[1] for g in groups), "node_attr0": node_attr.clone()} if return_intermediates else None
        L = len(self.conv_layers)
        for l, layer in enumerate(self.conv_layers):
            ng = 9 if l < L - 1 else 3
            e = cuts[ng]
            ea = torch.cat([edge_attr[:e], node_attr[edge_index[0, :e], :ns], node_attr[edge_index[1, :e], :ns]], -1)
            if c.differentiate_convolutions:
                ea = [ea[cuts[i]:cuts[i + 1]] for i in range(ng)]
            node_attr = layer(node_attr, edge_index[:, :e], ea, edge_sh[:e], edge_weight=edge_weight[:e])
            if inter is not None:
                inter[f"node_attr{l + 1}"] = node_attr.clone()
        if c.confidence_mode:   # aa_model.py:431-452 (atom_confidence=False, parallel=1)
            out = self._confidence(data, node_attr[:n_lig])
            return out + (inter,) if return_intermediates else out
        return self._readouts(data, node_attr[:n_lig], tr_sigma, rot_sigma, tor_sigma, inter, return_intermediates)
