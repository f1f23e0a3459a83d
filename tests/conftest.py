import os
import sys

# thread pools sized after the visible hardware threads (256 on the GPU box) spin a container's CPU quota (16 cores there)
# away and get the whole process parked: a sane default before numpy / torch start theirs (see gnnrag_amd.install)
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, str(max(1, min(16, os.cpu_count() or 1))))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's torch ops: no more OpenMP threads than the container's CPU quota allows (the GPU box shows 256
    # hardware threads and grants 16 cores; the surplus threads only get every thread of the process parked)
    try:
        import torch
        import gnnrag_amd  # noqa: F401
        from gnnrag_amd.install import host_cpu_budget
        torch.set_num_threads(max(1, min(torch.get_num_threads(), host_cpu_budget())))
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """npz fixture -> (cfg, batch, feats, params, ref) in the package's host types."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import synth
    z = np.load(os.path.join(GOLDEN, name))
    cfg = synth.GraphConfig(
        name=name, B=int(z["B"]), N=int(z["N"]), E=0, R=int(z["R1"]) - 2, D=int(z["D"]),
        I=int(z["I"]), L=int(z["L"]), T=int(z["T"]),
        normalized_gnn=bool(int(z["normalized_gnn"])), pos_emb=bool(int(z["pos_emb"])))
    F = len(z["heads"])
    edge_tuple = (z["heads"], z["rels"], z["tails"], z["batch_ids"], np.arange(F, dtype=np.int64),
                  z["weight_list"].tolist(), z["weight_rel_list"].tolist())
    batch = synth.Batch(cfg=cfg, local_entity=z["local_entity"], query_entities=z["query_entities"],
                        seed_dist=z["seed_dist"], edge_tuple=edge_tuple,
                        num_entity=int(z["num_entity"]), n_real=z["n_real"])
    feats = {k[5:]: z[k] for k in z.files if k.startswith("feat.")}
    params = {k[6:]: z[k] for k in z.files if k.startswith("param.")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref.")}
    return cfg, batch, feats, params, ref


@pytest.fixture(scope="session")
def golden_loader():
    return load_golden
