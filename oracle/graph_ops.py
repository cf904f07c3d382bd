"""Oracle restatement of the torch-cluster / torch-scatter ops on the path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Third-party, not vendored in
the reference: torch-cluster==1.6.0 (requirements.txt:19), torch-scatter==2.1.0
(requirements.txt:21).  Call sites: models/cg_model.py:477 (radius_graph),
:543-548 (cross radius, per-graph cutoff via scaled coordinates), :630 (bond
graph radius, default cap 32); models/tensor_layers.py:144,220 (scatter).

Neighbour-cap behaviour: torch-cluster's CUDA kernel scans x in ascending index
inside the query's batch segment and stops after max_num_neighbors hits; the CPU
path's order is KD-tree defined.  The oracle defines "first max_num_neighbors by
ascending x index" (the CUDA behaviour) -- parity unpinned at this boundary; the
synthetic ligands stay under the cap so it never binds in the parity tests.
"""
import torch


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    """For each y find x with same batch id and |x-y|^2 < r^2 (strict).
    Returns int64 [2, E]: row 0 = index into y, row 1 = index into x; grouped by
    ascending y, ascending x inside, capped per y."""
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    d2 = ((y[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    ok = (d2 < r * r) & (batch_y[:, None] == batch_x[None, :])
    if max_num_neighbors < x.shape[0]:
        rank = torch.cumsum(ok.long(), dim=1)
        ok = ok & (rank <= max_num_neighbors)
    yi, xi = torch.nonzero(ok, as_tuple=True)
    return torch.stack([yi, xi], 0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    """torch_cluster.radius_graph: radius(x, x, r, cap + (0 if loop else 1)), then
    row, col = e[1], e[0] for 'source_to_target'; self loops dropped."""
    e = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    row, col = (e[1], e[0]) if flow == "source_to_target" else (e[0], e[1])
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col], 0)


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """torch_scatter.scatter for dim=0, reduce in {'sum','mean'}."""
    assert dim == 0
    n = int(dim_size) if dim_size is not None else int(index.max()) + 1
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    if reduce == "mean":
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
        out = out / cnt.reshape(-1, *([1] * (src.dim() - 1)))
    elif reduce not in ("sum", "add"):
        raise NotImplementedError(reduce)
    return out


def scatter_mean(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim, dim_size, "mean")
