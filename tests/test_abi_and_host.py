"""CPU-side checks: the C-ABI library builds/loads and exports every symbol the header
declares, the host mirror keeps the reference's surface, and the product path refuses to
run without the GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden

import gnnrag_amd  # noqa: F401
from gnnrag_amd import _lib, synth


@pytest.fixture(scope="module")
def lib():
    from gnnrag_amd import build
    build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _header_functions():
    src = open(os.path.join(REPO, "include", "gnnrag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gnnrag_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = _header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
        assert n in _lib.SIGNATURES, "binding lacks a signature for " + n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.gnnrag_abi_version() == _lib.ABI_VERSION == 16
    assert b"bad argument" in lib.gnnrag_error_string(-1)


def test_size_queries_and_struct_layout(lib):
    assert ctypes.sizeof(_lib.CsrStruct) == 4 * 4 + 8 + 8 * 2 * 7 + 8 * 2 + 8 + 8 * 2 + 8 + 8 * 2 + 8 * 2 + 8 + 8 * 3 + 8 * 4
    n = lib.gnnrag_csr_bytes(768000, 64, 2000, 602, 0, 0)
    # 2 row_ptr arrays + 2 x (edge 8 B + compact-relation edge 8 B + perm 4 B) per fact, plus small
    # lists + one int per node for the per-question big-node lists + (question, relation) rows
    lo = 3 * 128001 * 4 + 768000 * 72 + 64 * 602 * 8          # (+ 32 B per fact: the merged record stream)
    assert lo <= n <= lo + 64 * 1024
    assert lib.gnnrag_csr_bytes(768000, 64, 2000, 602, 1, 1) >= n + 4 * 768000 * 4
    assert lib.gnnrag_csr_bytes(-1, 64, 2000, 602, 0, 0) == 0
    assert lib.gnnrag_csr_scratch_bytes(768000, 64, 2000, 602) >= 768000 * 4 + 64 * 602 * 4
    c = _lib.CsrStruct()
    c.B, c.N, c.R1, c.F, c.max_chunks = 64, 2000, 602, 768000, 2 * 3001
    c.rel_total, c.rel_max = 64 * 601, 601
    ws = lib.gnnrag_layer_workspace_bytes(ctypes.byref(c), 200, 2)
    # T tables + the larger of {agg [BN,2I*D]} / {P [2,rel_total,D] + nbr [BN,D]} + heavy-chunk partials
    assert ws >= 2 * 602 * 200 * 4 + 128000 * 800 * 4 + 2 * c.max_chunks * 2 * 200 * 4
    assert lib.gnnrag_aggregate_workspace_bytes(ctypes.byref(c), 200, 2) >= 2 * c.max_chunks * 2 * 200 * 4
    # the stack workspace adds all L layers' relation projections [L,2,R1,D] and, at the V-form table kernel's hidden
    # sizes, their bf16 planes [L,2,3,R1,448]
    planes = lib.gnnrag_rel_planes_bytes(602, 200, 3)
    assert planes == 3 * 2 * 3 * 602 * 448 * 2
    assert lib.gnnrag_stack_workspace_bytes(ctypes.byref(c), 3, 200, 2) >= ws + 3 * 2 * 602 * 200 * 4 + planes
    assert lib.gnnrag_stack_workspace_bytes(ctypes.byref(c), 3, 64, 2) >= lib.gnnrag_layer_workspace_bytes(ctypes.byref(c), 64, 2) + 3 * 2 * 602 * 64 * 4
    assert lib.gnnrag_rel_planes_bytes(602, 300, 3) == 0 and lib.gnnrag_stack_workspace_bytes(None, 3, 200, 2) == 0
    # weight-gradient GEMM: partial blocks of [N1, N2] per row chunk
    tn = lib.gnnrag_gemm_tn_workspace_bytes(128000, 200, 200)
    assert tn >= 200 * 200 * 4 and tn % (200 * 200 * 4) == 0 and lib.gnnrag_gemm_tn_workspace_bytes(0, 200, 200) == 0


def test_argument_errors_without_gpu(lib):
    assert lib.gnnrag_masked_softmax(None, None, 1, 1, None) == -1
    assert lib.gnnrag_linear(None, 1, 1, None, None, None, 0, 0, None, 1, 0, None) == -1
    assert lib.gnnrag_aggregate(None, None, None, None, None, None, 200, 2, None, 0, None) == -1
    assert lib.gnnrag_relation_tables(None, None, None, None, None, None, 8, 2, 0, None) == -1
    # the math mode is an argument (no library-wide mode): unknown values are rejected before anything is launched
    one = ctypes.c_void_p(256)
    assert lib.gnnrag_linear(one, 1, 1, one, None, None, 0, 0, one, 1, 7, None) == -1
    c = _lib.CsrStruct()
    c.rel_total, c.rel_max = 10, 5
    assert lib.gnnrag_aggregate_fused_variant(ctypes.byref(c), 200) == 2         # small tables: 32-column LDS slices
    c.rel_max = 602
    assert lib.gnnrag_aggregate_fused_variant(ctypes.byref(c), 200) == 1
    c.rel_max = 6001
    assert lib.gnnrag_aggregate_fused_variant(ctypes.byref(c), 200) == 0         # tables exceed a CU's LDS
    assert lib.gnnrag_aggregate_fused_variant(None, 200) == -1
    # round-2 entry points: null pointers / bad sizes -> BADARG, shapes outside a kernel's set -> UNSUPPORTED (-2),
    # nothing is launched either way
    assert lib.gnnrag_rel_transform(None, None, 10, 8, 1, None, 0, None, None, None) == -1
    lp = (_lib.LayerParams * 1)()
    lp[0].W_rel, lp[0].b_rel = 256, 256
    assert lib.gnnrag_rel_transform(one, one, 10, 6, 1, lp, 0, one, None, None) == -2        # D % 4 != 0
    assert lib.gnnrag_rel_transform(one, one, 10, 300, 1, lp, 0, one, one, None) == -2      # planes need D <= 224
    assert lib.gnnrag_relation_tables_planes(None, None, None, None, None, 200, 2, None) == -1
    c.rel_total, c.rel_max = 100, 50
    assert lib.gnnrag_relation_tables_planes(ctypes.byref(c), one, one, one, one, 200, 2, None) == -2   # < 1024 rows
    assert lib.gnnrag_gemm_tn(None, None, 10, 8, 8, None, None, 0, None) == -1
    assert lib.gnnrag_gemm_tn(one, one, 10, 6, 8, one, one, 1 << 20, None) == -2            # N1 % 4 != 0
    assert lib.gnnrag_gemm_tn(one, one, 10, 8, 8, one, None, 0, None) == -3                 # workspace too small
    # path flags: one direction only, not both
    args = [ctypes.byref(c)] + [one] * 9 + [0] + [one] * 8 + [one, 1 << 30, 200, 2]
    assert lib.gnnrag_reason_layer(*args, 2 | 0x10 | 0x20, 2, None) == -1
    assert lib.gnnrag_reason_layer(*args, 5, 2, None) == -1


def test_module_surface_matches_reference_state_dict():
    """Parameter names/shapes are those recorded from the reference layer (fixture), so released
    checkpoints load (SURVEY.md section 5 'Checkpoint')."""
    from gnnrag_amd.modules.kg_reasoning.reasongnn import ReasonGNNLayer
    for name in ("layer_d200.npz", "layer_d50.npz"):
        cfg, batch, feats, params, _ = load_golden(name)
        args = dict(use_cuda=True, normalized_gnn=cfg.normalized_gnn, num_ins=cfg.I, num_gnn=cfg.L,
                    pos_emb=cfg.pos_emb, linear_dropout=0.0)
        layer = ReasonGNNLayer(args, batch.num_entity, cfg.num_kb_relation, cfg.D, "bfs")
        want = {k: v.shape for k, v in params.items() if not k.startswith("type_layer.")}
        got = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        assert got == want
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()
                               if not k.startswith("type_layer.")}, strict=True)
        for m in ("init_reason", "forward", "build_matrix", "init_layers"):
            assert callable(getattr(layer, m))


def test_product_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gnnrag_amd import ops, stack
    cfg = synth.CONFIGS["tiny"]
    batch = synth.make_batch(cfg)
    with pytest.raises(_lib.GnnragError):
        ops.CsrPlan(batch.edge_tuple[0], batch.edge_tuple[1], batch.edge_tuple[2], cfg.B, cfg.N, cfg.R1, "cpu")
    layer = stack.build_layer(cfg, batch, synth.make_layer_params(cfg), "cpu")
    with torch.no_grad(), pytest.raises(_lib.GnnragError):
        layer.init_reason(local_entity=torch.from_numpy(batch.local_entity), kb_adj_mat=batch.edge_tuple,
                          local_entity_emb=torch.zeros(cfg.B, cfg.N, cfg.D),
                          rel_features=torch.zeros(cfg.R1, cfg.D), rel_features_inv=torch.zeros(cfg.R1, cfg.D),
                          query_entities=torch.zeros(cfg.B, cfg.N))
    with pytest.raises(_lib.GnnragError):
        ops.linear(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(_lib.GnnragError):
        ops.gemm_tn(torch.zeros(8, 4), torch.zeros(8, 4))
    with pytest.raises(_lib.GnnragError):
        ops.rel_transform(torch.zeros(4, 8), torch.zeros(4, 8), [(torch.zeros(8, 8), torch.zeros(8), None, None)])
    from gnnrag_amd.autograd import linear as ag_linear
    with pytest.raises(_lib.GnnragError):                    # the training-path projections have no CPU form either
        ag_linear(torch.zeros(4, 8, requires_grad=True), torch.zeros(8, 8), torch.zeros(8))


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(REPO, "gnn-rag_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(root, f)


def test_install_substitutes_reference_module_names():
    import sys
    from gnnrag_amd import install
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "modules" or k.startswith("modules.")}
    try:
        install.install()
        from modules.kg_reasoning.reasongnn import ReasonGNNLayer     # noqa: the reference's import line
        from modules.layer_init import TypeLayer                       # noqa
        assert ReasonGNNLayer.__module__.startswith("gnnrag_amd.")
        assert TypeLayer.__module__.startswith("gnnrag_amd.")
        from modules.query_update import AttnEncoder, Fusion, QueryReform   # noqa: rearev.py:12
        assert all(c.__module__.startswith("gnnrag_amd.") for c in (AttnEncoder, Fusion, QueryReform))
        # parameter names of the reference classes (query_update.py:9-10,23-24,50)
        assert sorted(QueryReform(8).state_dict()) == ["fusion.g.weight", "fusion.r.weight", "q_ent_attn.bias",
                                                       "q_ent_attn.weight"]
        assert sorted(AttnEncoder(8).state_dict()) == ["attn_linear.weight"]
    finally:
        install.uninstall()
        sys.modules.update(saved)


def test_synth_edge_tuple_semantics():
    """make_edge_tuple restates _build_fact_mat (dataset_load.py:473-527): offsets, contiguous
    questions, self loops with rel id num_kb_relation-1, 1/outdeg and 1/count(head,rel)."""
    from collections import Counter
    cfg = synth.CONFIGS["tiny50"]
    batch = synth.make_batch(cfg)
    h, r, t, b, f, wl, wrl = batch.edge_tuple
    assert (np.diff(b) >= 0).all() and (h // cfg.N == b).all() and (t // cfg.N == b).all()
    loops = r == cfg.num_kb_relation - 1
    assert (h[loops] == t[loops]).all() and loops.sum() == batch.n_real.sum()
    hc = Counter(h.tolist())
    hrc = Counter(zip(h.tolist(), r.tolist()))
    assert wl == [1.0 / hc[x] for x in h.tolist()]
    assert wrl == [1.0 / hrc[x] for x in zip(h.tolist(), r.tolist())]
    assert (f == np.arange(len(h))).all()


def test_integration_stub_struct_matches_the_binding():
    """The ctypes stub INTEGRATION.md shows to a maintainer lists the fields of ``struct gnnrag_csr`` in the order
    and with the types of the binding the product uses (it went stale once: two ABI versions behind)."""
    import re
    from gnnrag_amd import _lib
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = text[text.index("class Csr(C.Structure)"):]
    block = block[:block.index("]")]
    got = [(n, getattr(ctypes, t), bool(a)) for n, t, a in re.findall(r'\("(\w+)", C\.(c_\w+)( \* 2)?\)', block)]
    want = []
    for name, ctype in _lib.CsrStruct._fields_:
        arr = hasattr(ctype, "_length_")
        want.append((name, ctype._type_ if arr else ctype, arr))
    assert got == want


def test_pmc_source_digest_ignores_comments_and_whitespace(tmp_path, monkeypatch):
    """bench.kernel_sources_digest stamps profiles/pmc_traffic.json: a comment edit must not declare the measured
    traffic stale, a code edit must."""
    import shutil
    import bench
    src = os.path.join(REPO, "gnn-rag_amd", "csrc")
    dst = tmp_path / "gnn-rag_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in ("aggregate.hip", "csr_plan.hip", "frontier.hip", "gnnrag_common.h"):
        shutil.copy(os.path.join(src, f), dst / f)
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    base = bench.kernel_sources_digest()
    p = dst / "frontier.hip"
    text = p.read_text()
    p.write_text("// a new remark\n" + text.replace("\n", "\n   \n", 3) + "/* and a block\n comment */\n")
    assert bench.kernel_sources_digest() == base
    p.write_text(text.replace("kFrAltSeeds = 32", "kFrAltSeeds = 16"))
    assert bench.kernel_sources_digest() != base


def test_host_cpu_budget_follows_the_cgroup_quota(tmp_path):
    """install.host_cpu_budget: the CFS quota of cgroup v2 (cpu.max) or v1 (cfs_quota_us / cfs_period_us) caps
    os.cpu_count(); "max" / -1 / no cgroup files leave it alone.  (The MI355X box: 256 visible hardware threads, 16 granted -
    torch's default thread count follows the former and gets the whole process parked, profiles/r05i_forward_host_time.txt.)"""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd.install import host_cpu_budget
    ncpu = os.cpu_count() or 1
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("200000 100000\n")
    assert host_cpu_budget(str(v2)) == min(ncpu, 2)
    (v2 / "cpu.max").write_text("50000 100000\n")            # half a core: at least one thread
    assert host_cpu_budget(str(v2)) == 1
    (v2 / "cpu.max").write_text("max 100000\n")
    assert host_cpu_budget(str(v2)) == ncpu
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert host_cpu_budget(str(v1)) == min(ncpu, 3)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert host_cpu_budget(str(v1)) == ncpu
    assert host_cpu_budget(str(tmp_path / "nothing_here")) == ncpu


def test_limit_host_threads_sets_a_small_team():
    import torch
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd.install import host_cpu_budget, host_thread_limit, limit_host_threads
    before = torch.get_num_threads()
    try:
        n = limit_host_threads()
        assert n == torch.get_num_threads() == host_thread_limit() <= max(1, min(8, host_cpu_budget() // 2))
        assert limit_host_threads(3) == torch.get_num_threads() == 3
    finally:
        torch.set_num_threads(before)


def test_host_thread_limit_divides_the_quota_by_the_local_ranks():
    """the ranks of one node share ONE CPU quota - 8 ranks on a 16-core quota get 1 thread each, not 8;
    an OMP_NUM_THREADS the launcher set (torch.distributed.run: 1) is never exceeded."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd.install import host_thread_limit
    assert host_thread_limit(16, {}) == 8
    assert host_thread_limit(16, {"WORLD_SIZE": "8"}) == 1
    assert host_thread_limit(16, {"WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "2"}) == 4
    assert host_thread_limit(64, {"LOCAL_WORLD_SIZE": "8"}) == 4
    assert host_thread_limit(64, {"LOCAL_WORLD_SIZE": "8", "OMP_NUM_THREADS": "1"}) == 1
    assert host_thread_limit(64, {"OMP_NUM_THREADS": "12"}) == 8
    assert host_thread_limit(1, {"WORLD_SIZE": "8"}) == 1
    assert host_thread_limit(16, {"WORLD_SIZE": "junk", "OMP_NUM_THREADS": ""}) == 8


def test_bench_limits_the_thread_pools_before_importing_numpy_and_torch():
    """bench.py sets OMP / OpenBLAS / MKL pool sizes before numpy and torch are imported (pools sized after the visible
    hardware threads spin a container's CPU quota away: profiles/r05n_host_stall_in_timed_region.txt); an explicit setting
    wins, GNNRAG_HOST_THREADS=0 switches the rule off."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import bench; "
            "print(os.environ.get('OMP_NUM_THREADS'), os.environ.get('OPENBLAS_NUM_THREADS'), os.environ.get('MKL_NUM_THREADS'))" % REPO)
    base = {k: v for k, v in os.environ.items()
            if k not in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "GNNRAG_HOST_THREADS")}

    def run(extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.strip().splitlines()[-1].split()

    got = run({})
    assert all(v.isdigit() and 1 <= int(v) <= 8 for v in got), got
    assert run({"OMP_NUM_THREADS": "3"})[0] == "3"
    # eight local ranks share the quota: this container's 8 cores // (2 * 8) -> 1 thread per pool
    assert run({"WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}) == ["1", "1", "1"] or (os.cpu_count() or 1) >= 32
    assert run({"GNNRAG_HOST_THREADS": "0"}) == ["None", "None", "None"]


def test_cache_rel_features_reuses_until_a_parameter_changes():
    """install.cache_rel_features: ``ReaRev.get_rel_feature`` (rearev.py:91-111) re-encodes the whole relation vocabulary
    on every forward; in evaluation (eval mode, no grad) the result is kept until a model parameter changes (storage or
    version), training / grad-enabled calls always recompute."""
    import torch
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd.install import cache_rel_features

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.relation_embedding = torch.nn.Embedding(7, 4)
            self.relation_linear = torch.nn.Linear(4, 3)
            self.calls = 0

        def get_rel_feature(self):
            self.calls += 1
            return self.relation_linear(self.relation_embedding.weight), self.relation_linear(self.relation_embedding.weight)

    m = cache_rel_features(M()).eval()
    assert cache_rel_features(m) is m                       # idempotent
    with torch.no_grad():
        a = m.get_rel_feature()
        b = m.get_rel_feature()
        assert m.calls == 1 and a[0] is b[0]
        m.relation_linear.weight.add_(1.0)                   # in-place update: the version counter moves
        c = m.get_rel_feature()
        assert m.calls == 2 and not torch.equal(c[0], a[0])
        m.load_state_dict({k: v.clone() * 0.5 for k, v in m.state_dict().items()})       # a checkpoint load
        m.get_rel_feature()
        assert m.calls == 3
    m.get_rel_feature()                                      # grad enabled: never cached
    m.get_rel_feature()
    assert m.calls == 5
    m.train()
    with torch.no_grad():
        m.get_rel_feature()
        m.get_rel_feature()
    assert m.calls == 7
