"""CPU oracle for the DiffDock score-model sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``diffdock_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and there only as the checker / reported baseline.

What it is: a pure-torch (CPU, fp32 or fp64) restatement of the reference path
(``/root/reference`` = gcorso/DiffDock @ 2024_10_08), file by file:

  oracle/e3nn_lite.py   e3nn==0.5.x pieces the path uses (NOT vendored in the
                        reference): Irreps, wigner_3j, spherical_harmonics,
                        FullyConnectedTensorProduct, FullTensorProduct, BatchNorm,
                        o3.Linear and the 'uvu' o3.TensorProduct (round 5: sidechain_pred,
                        depthwise convolutions)
  oracle/graph_ops.py   torch-cluster 1.6 radius / radius_graph, torch-scatter scatter
  oracle/layers.py      models/layers.py + models/tensor_layers.py
  oracle/cg_model.py    models/cg_model.py (CGModel forward)
  oracle/conformer.py   utils/geometry.py, utils/torsion.py, utils/diffusion_utils.py
  oracle/tables.py      utils/so3.py, utils/torus.py score-norm tables
  oracle/sampling.py    utils/sampling.py reverse-diffusion loop

Parity pinning: the reference has no tests / golden vectors of its own and its
third-party deps (e3nn, torch_scatter, torch_cluster, torch_geometric) are absent,
so the pins are fixtures produced by EXECUTING the reference's own python
(models/cg_model.py, models/tensor_layers.py, models/layers.py, utils/geometry.py,
utils/torsion.py, utils/diffusion_utils.py, utils/sampling.py, utils/so3.py,
utils/torus.py) with only the absent third-party modules substituted
(tests/golden/make_golden.py and make_golden_fullsize.py -- the latter at the BASELINE.json
shapes: 20 steps x 10 poses at 300 residues / 30 atoms, one forward at 1500 / 80 --
fixtures committed under tests/golden/).  The
third-party arithmetic itself (e3nn w3j / SH / BatchNorm, radius cap order) is a
restatement of the published algorithm and stays "parity unpinned" at that
boundary except for the in-repo pin FasterTensorProduct
(models/tensor_layers.py:44-122), which fixes all l<=1 constants.
"""
