"""tests/golden/1a0q_graph.pt: the reference's example complex data/1a0q (416 residues, 23 heavy atoms; BASELINE configs[0]).

    python tests/golden/make_1a0q.py        (THIS container: the GPU box has no /root/reference)

What is REFERENCE-EXECUTED in this fixture (the reference's own functions, unmodified, run under make_golden.install_stubs()):
  * datasets/process_mols.py:161-200 `new_extract_receptor_structure` -- the C-alpha neighbour graph (torch.cdist distances in
    float32 on the UNCENTRED coordinates, cutoff 15 A, the 24 nearest when there are more), residue-type feature column,
    receptor positions -- fed with the residue sequence and N / CA / C coordinates parsed from the PDB file
    (`get_chi_angles`, RDKit/ProDy-free numpy in the reference but irrelevant to the path, is stubbed to zeros:
    side_chain_vecs are never read by the score model);
  * utils/torsion.py:15-45 `get_transformation_mask` -- rotatable-bond edge mask and atom masks -- on the ligand bond
    graph parsed from the SDF file, with a networkx stand-in for torch_geometric.utils.to_networkx;
  * the centring of datasets/pdbbind.py:405-416 (float32 mean of the receptor positions).
What stays an external input (third-party perception that cannot run here): PDB / SDF parsing itself (ProDy, RDKit --
done by diffdock_amd.io's readers), the 15 RDKit chemistry features of a ligand atom (only the atomic-number column is
filled), ESM embeddings (lm_dim = 0).  tests/test_io.py compares diffdock_amd.io.complex_graph with this fixture."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from diffdock_amd import io as dio  # noqa: E402
from diffdock_amd.hetero import HeteroData  # noqa: E402
from make_golden import graph_to_dict, install_stubs  # noqa: E402

D = "/root/reference/data/1a0q"
THREE2ONE = {'ALA': 'A', 'ARG': 'R', 'ASN': 'N', 'ASP': 'D', 'CYS': 'C', 'GLN': 'Q', 'GLU': 'E', 'GLY': 'G', 'HIS': 'H', 'ILE': 'I',
             'LEU': 'L', 'LYS': 'K', 'MET': 'M', 'PHE': 'F', 'PRO': 'P', 'SER': 'S', 'THR': 'T', 'TRP': 'W', 'TYR': 'Y', 'VAL': 'V'}


def main():
    scratch = os.path.join(ROOT, ".scratch", "tables")
    os.makedirs(scratch, exist_ok=True)
    os.chdir(scratch)
    install_stubs()
    import networkx as nx
    import datasets.process_mols as pm
    import utils.torsion as tz

    # ---- receptor: sequence + backbone coordinates from the file, graph / features / positions by the reference
    backbone, names = dio.read_pdb_backbone(f"{D}/1a0q_protein_processed.pdb")          # [n, 3 (N, CA, C), 3] float64, residue names
    seq = "".join(THREE2ONE.get(n, "X") for n in names)                                  # ProDy's pdb.ca.getSequence()
    pm.get_chi_angles = lambda coords, seq_, return_onehot=True: (np.zeros((len(seq_), 4)), None)
    g = HeteroData()
    pm.new_extract_receptor_structure(seq, backbone, g, neighbor_cutoff=15.0, max_neighbors=24, lm_embeddings=None,
                                      knn_only_graph=False, all_atoms=False)
    g["receptor"].side_chain_vecs = torch.zeros(len(seq), 10)

    # ---- ligand: atoms / bonds from the file, rotatable-bond masks by the reference
    lc, z, bonds = dio.read_sdf(f"{D}/1a0q_ligand.sdf")
    ei, attr = dio.ligand_bond_arrays(bonds)
    feats = np.zeros((len(z), 16), dtype=np.int64)
    feats[:, 0] = np.where((z >= 1) & (z <= 118), z - 1, 118)
    g["ligand"].x = torch.from_numpy(feats)
    g["ligand"].pos = torch.from_numpy(lc.astype(np.float32))
    g["ligand", "lig_bond", "ligand"].edge_index = torch.from_numpy(ei)
    g["ligand", "lig_bond", "ligand"].edge_attr = torch.from_numpy(attr)

    class Homogeneous:          # what to_networkx(pyg_data.to_homogeneous(), to_undirected=False) sees: ligand nodes and bonds
        num_nodes, edge_index = len(z), torch.from_numpy(ei)

    def to_networkx(data, to_undirected=False):
        G = nx.DiGraph()
        G.add_nodes_from(range(data.num_nodes))
        G.add_edges_from((int(a), int(b)) for a, b in data.edge_index.T.tolist())
        return G
    tz.to_networkx = to_networkx

    class View:                 # the two accesses get_transformation_mask makes on the complex graph
        def to_homogeneous(self):
            return Homogeneous()

        def __getitem__(self, k):
            return g[k]
    view = View()
    mask_edges, mask_rotate = tz.get_transformation_mask(view)
    g["ligand"].edge_mask = torch.as_tensor(mask_edges)
    g["ligand"].mask_rotate = [np.asarray(mask_rotate)]

    # ---- datasets/pdbbind.py:405-416
    center = torch.mean(g["receptor"].pos, dim=0, keepdim=True)
    g["receptor"].pos -= center
    g["ligand"].pos -= center
    g.original_center = center
    d = graph_to_dict(g)
    d["original_center"] = center
    d["provenance"] = "receptor graph / features / positions: datasets/process_mols.py new_extract_receptor_structure executed; " \
                      "edge_mask / mask_rotate: utils/torsion.py get_transformation_mask executed; file parsing: diffdock_amd.io"
    torch.save(d, os.path.join(HERE, "1a0q_graph.pt"))
    print({k: tuple(v.shape) for k, v in d.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
