#!/usr/bin/env python3
"""bench.py -- snippets/sec of one training step of the Snipper hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its own N ranks as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric` / configs[2..3]): T=4 frames of 600x800 per snippet, ResNet-50
backbone -> 3 feature levels -> hidden_dim 384, 8 heads, enc6/dec6, 60 queries, batch 2 snippets per
GPU, synthetic images, random-init weights.  Precision: bf16 autocast for the dense layers (BASELINE
configs[2] names bf16; the reference itself has no AMP, `--precision fp32` reproduces that), fp32 master
weights and optimizer, fp32 deformable-attention sampling.  One step = forward,
loss, backward, gradient clipping (0.1, engine.py:74) and an AdamW update.  One process per GPU;
gradients are all-reduced by DistributedDataParallel over RCCL, overlapped with backward
(weak scaling: every rank keeps its own 2 snippets per step).

The loss is the reference's SetCriterion with its Hungarian matcher (snipper_amd/criterion.py mirrors
models/model.py:240-545 and models/matcher.py, all decoder layers per call, one host round trip per step) on
synthetic targets in the dataloader's format; `--loss surrogate` is the earlier fixed-assignment stand-in.

One JSON line on stdout (rank 0).  Besides the contract's keys:
  roofline      the dominant hand-written kernel (deformable-attention backward, encoder shape),
                timed live with events on the launch stream, against the 8 TB/s HBM peak with the
                algorithmic bytes of SURVEY.md section 8(d) / DESIGN.md;
  cpu_baseline  the CPU oracle's use_pytorch_deform=1 formulation on the host's physical cores: one encoder core
                call, one encoder MSDeformAttn module fwd+bwd (= `value`, ms) and BASELINE configs[0] (T=1 enc2/dec4
                transformer forward); measured, nothing extrapolated;
  msda          MSDeformAttn fwd+bwd ms (encoder / decoder module), the second half of the metric.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn.functional as F

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same table; AMD's 5 PF figure includes 2:1 sparsity)


def model_args(a):
    return SimpleNamespace(
        hidden_dim=a.hidden_dim, nheads=8, enc_layers=a.enc_layers, dec_layers=a.dec_layers, dim_feedforward=1024,
        dropout=0.1, num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=a.frames,
        num_future_frames=a.future_frames, use_pytorch_deform=bool(a.use_pytorch_deform), num_kpts=15,
        position_embedding="sine", backbone="resnet50", lr_backbone=1e-5, masks=False, dilation=False,
        num_queries=60, aux_loss=True)


def surrogate_loss(out, tgt):
    """Fixed assignment: query i <-> synthetic person i for i < m; the rest are background.  Summed over the
    decoder layers (main + auxiliary outputs, like the reference's aux_loss), plus a small heat-map term."""
    m = tgt["kpts2d"].shape[1]
    al = out.get("all_layers")
    if al is not None:      # all decoder layers in one shot: sum_l mean_l(x) == n_layers * mean(x) for equal shapes
        logits, kpts = al["pred_logits"], al["pred_kpts"]                # [n_dec, bs, nq, t, 2], [n_dec, bs, nq, t, K, 4]
        n_dec = logits.shape[0]
        labels = tgt["labels"].unsqueeze(0).expand(n_dec, -1, -1, -1)
        total = F.cross_entropy(logits.flatten(0, 3).float(), labels.reshape(-1))
        total = total + F.l1_loss(kpts[:, :, :m, ..., 0:3].float(), tgt["kpts2d"].unsqueeze(0).expand(n_dec, -1, -1, -1, -1, -1)) * 5.0
        total = total + F.l1_loss(kpts[:, :, :m, ..., 3:4].float(), tgt["depth"].unsqueeze(0).expand(n_dec, -1, -1, -1, -1, -1))
        total = total * n_dec
    else:
        def one(o):
            loss = F.cross_entropy(o["pred_logits"].flatten(0, 2), tgt["labels"].flatten())
            loss = loss + F.l1_loss(o["pred_kpts2d"][:, :m], tgt["kpts2d"]) * 5.0
            return loss + F.l1_loss(o["pred_depth"][:, :m], tgt["depth"])
        total = one(out)
        for aux in out.get("aux_outputs", []):
            total = total + one(aux)
    for hm in out["heatmaps"]:
        total = total + hm.pow(2).mean() * 0.01
    return total


def make_batches(a, device, n_batches, seed):
    """Synthetic snippets in the reference dataloader's item format (datasets/hybrid_dataloader.py:1072-1098):
    images [T*3, H, W] in [0,1]; per snippet a target dict with m persons: kpts2d [m, T+F, 15, 3] (x, y in [0,1],
    visibility, 80 % visible), depth [m, T+F, 15, 2] (value, exists), traj_ids [m], max_depth."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t_all = a.frames + a.future_frames
    batches = []
    for _ in range(n_batches):
        imgs = torch.rand(a.batch, a.frames * 3, a.height, a.width, generator=g).to(device)
        targets = []
        for i in range(a.batch):
            m = 6 + 2 * i                                   # 6, 8, ... persons
            k2 = torch.rand(m, t_all, 15, 3, generator=g)
            k2[..., 2] = (torch.rand(m, t_all, 15, generator=g) < 0.8).float()
            d = torch.rand(m, t_all, 15, 2, generator=g)
            d[..., 1] = (torch.rand(m, t_all, 15, generator=g) < 0.8).float()
            targets.append({"kpts2d": k2.to(device), "depth": d.to(device), "traj_ids": torch.arange(m, device=device),
                            "max_depth": torch.tensor(15.0, device=device)})
        labels = torch.zeros(a.batch, 60, t_all, dtype=torch.long)
        labels[:, :8] = 1
        tgt = {"targets": targets, "labels": labels.to(device),          # the last three feed the surrogate loss only
               "kpts2d": torch.stack([t["kpts2d"][:6] for t in targets]),
               "depth": torch.stack([t["depth"][:6, ..., :1] for t in targets])}
        batches.append((imgs, tgt))
    return batches


def criterion_args(a):
    """The loss / matcher coefficients of the reference's CLI defaults (main.py:109-146)."""
    return SimpleNamespace(
        max_depth=15, set_cost_is_human=1, set_cost_root=1, set_cost_root_depth=1, set_cost_root_vis=0.1,
        set_cost_joint=1, set_cost_joint_depth=1, set_cost_joint_vis=0.1, is_human_loss_coef=1, root_loss_coef=1,
        root_depth_loss_coef=1, root_vis_loss_coef=0.1, joint_loss_coef=1, joint_depth_loss_coef=1, joint_vis_loss_coef=1,
        joint_disp_loss_coef=1, joint_disp_depth_loss_coef=1, cont_loss_coef=0.1, heatmap_loss_coef=0.01, eos_coef=0.5,
        aux_loss=True, dec_layers=a.dec_layers)


def decoder_side(name: str) -> bool:
    """Parameters whose gradients are complete once the decoder's backward is (the first ~20 % of backward): decoder layers,
    the heads inside it, the query / joint embeddings."""
    return name.startswith(("transformer.decoder.", "query_embed", "joint_embed", "transformer.reference_points"))


def _encoder_layer_index(name: str):
    """i for "transformer.encoder.layers.<i>....", else None."""
    pre = "transformer.encoder.layers."
    if not name.startswith(pre):
        return None
    head = name[len(pre):].split(".", 1)[0]
    return int(head) if head.isdigit() else None


def _encoder_upper_from(named_params) -> int:
    """First layer of the encoder's UPPER half (its gradients are complete first: backward walks the layers downwards)."""
    idx = [i for i in (_encoder_layer_index(n) for n, _ in named_params) if i is not None]
    return (max(idx) + 1 + 1) // 2 if idx else 0


def optimizer_groups(named_params):
    """The reference's three groups (main.py:201-221) over (name, parameter) pairs: (main, backbone, slow) lists.  Inside
    `main` the parameters are listed in the order backward COMPLETES them, as a stable partition into three runs -- decoder
    side, upper half of the encoder's layers, everything else -- so that each run is contiguous in the flat layout and the
    gradient all-reduce can launch it as a stage of its own (grad_sync_stages).  The optimizer's state dict does not see
    this order (FlatAdamW maps ids through reference_param_groups)."""
    named_params = list(named_params)

    def named(pred):
        return [p for n, p in named_params if p.requires_grad and pred(n)]
    slow = lambda n: ("reference_points" in n or "sampling_offsets" in n) and "backbone" not in n
    is_main = lambda n: "backbone" not in n and not slow(n)
    upper_from = _encoder_upper_from(named_params)
    upper = lambda n: (_encoder_layer_index(n) or 0) >= upper_from and _encoder_layer_index(n) is not None and upper_from > 0
    main = (named(lambda n: is_main(n) and decoder_side(n)) +
            named(lambda n: is_main(n) and not decoder_side(n) and upper(n)) +
            named(lambda n: is_main(n) and not decoder_side(n) and not upper(n)))
    return (main, named(lambda n: "backbone" in n), named(slow))


def reference_param_groups(named_params):
    """The reference optimizer's param_groups exactly as main.py:201-217 builds them: (main, backbone, slow), each in
    ``model.named_parameters()`` order.  ``checkpoint['optimizer']`` numbers parameters by position in THESE lists, so
    FlatAdamW maps its state dict through them (``reference_groups``), whatever order the flat layout uses."""
    match = lambda n, kws: any(k in n for k in kws)
    bb, slow = ["backbone.0"], ["reference_points", "sampling_offsets"]
    named_params = [(n, p) for n, p in named_params if p.requires_grad]
    return ([p for n, p in named_params if not match(n, bb) and not match(n, slow)],
            [p for n, p in named_params if match(n, bb)],
            [p for n, p in named_params if match(n, slow)])


def grad_sync_stages(model, non_backbone, names_out=None):
    """Stages of the gradient all-reduce, in the order backward completes them (round 5: eight, none above 43 MB):
      1. the decoder side of the `main` group (complete when the first decoder layer's and the query embedding's gradients
         are: ~30 % into backward);
      2. the upper half of the encoder's layers (complete when the LOWEST of them has run backwards);
      3. the rest of everything but the backbone: lower encoder layers, input projections, embeddings, the slow group
         (complete when the 1x1 input projections' gradients are);
      4-6. layer4 of the ResNet block by block (its three bottlenecks run backwards 2, 1, 0), 7. layer3, 8. layer2 (each
         complete when its FIRST block's gradients are: the blocks of a stage run backwards).
    Each stage's slice is all-reduced over RCCL while backward is still working on the next one.  (Round 3 had 1-3 as ONE
    70 MB stage launched 80 % into backward; round 4 had 2 + 3 as one 36 MB stage complete at 14.5 of 18 ms and layer4 as
    one 60 MB stage at 15.5 ms: 129 of 171 MB entered the wire in the last 20 % of backward -- grad_sync_trace.)
    ``non_backbone`` = main + slow in flat-layout order (optimizer_groups); ``names_out``: a list that receives one label
    per stage."""
    names = {id(p): n for n, p in model.named_parameters()}
    labels = []
    n_dec = 0
    while (n_dec < len(non_backbone) and decoder_side(names[id(non_backbone[n_dec])]) and
           "sampling_offsets" not in names[id(non_backbone[n_dec])] and "reference_points" not in names[id(non_backbone[n_dec])]):
        n_dec += 1
    upper_from = _encoder_upper_from(list(model.named_parameters()))
    n_up = n_dec
    while (n_up < len(non_backbone) and upper_from > 0 and
           (_encoder_layer_index(names[id(non_backbone[n_up])]) or -1) >= upper_from):
        n_up += 1
    stages = []
    rest_from = 0
    if 0 < n_dec < len(non_backbone):
        head = {id(p) for p in non_backbone[:n_dec]}
        dec0 = [p for n, p in model.named_parameters() if p.requires_grad and id(p) in head and
                (n.startswith("transformer.decoder.layers.0.") or n.startswith("query_embed"))]
        stages.append((non_backbone[:n_dec], dec0))
        labels.append("decoder + heads + queries")
        rest_from = n_dec
        if n_dec < n_up < len(non_backbone):
            run = non_backbone[n_dec:n_up]
            trig = [p for p in run if _encoder_layer_index(names[id(p)]) == upper_from]
            stages.append((run, trig))
            labels.append(f"encoder layers {upper_from}..")
            rest_from = n_up
    stages.append((non_backbone[rest_from:], list(model.input_proj.parameters())))
    labels.append("lower encoder layers + input projections + slow group" if rest_from > n_dec else
                  ("encoder + input projections + slow group" if rest_from else "everything but the backbone"))
    body = model.backbone[0].body
    for name in ("layer4", "layer3", "layer2"):
        layer = getattr(body, name, None)
        if layer is None:
            continue
        blocks = list(layer) if name == "layer4" else None
        if blocks and len(blocks) > 1 and all(any(p.requires_grad for p in b.parameters()) for b in blocks):
            for bi in range(len(blocks) - 1, -1, -1):          # a stage per bottleneck, in backward order
                ps = [p for p in blocks[bi].parameters() if p.requires_grad]
                stages.append((ps, ps))
                labels.append(f"{name}.{bi}")
            continue
        ps = [p for p in layer.parameters() if p.requires_grad]
        if ps:
            stages.append((ps, [p for p in layer[0].parameters() if p.requires_grad]))
            labels.append(name)
    if names_out is not None:
        names_out[:] = labels
    return stages


def build_optimizer(named_params, capturable=False, flat=None):
    """AdamW with the reference's three groups (main.py:201-221).  ``flat``: a FlatParameters over (main, slow,
    backbone) -- then each group is its one flat leaf."""
    main, backbone, slow = optimizer_groups(named_params)
    if flat is not None:
        main, slow, backbone = ([flat.leaf_of_group(i)] if flat.leaf_of_group(i) is not None else [] for i in range(3))
    groups = [
        {"params": main, "lr": 1e-4},
        {"params": backbone, "lr": 1e-5},
        {"params": slow, "lr": 1e-5},
    ]
    groups = [g for g in groups if g["params"]]
    # fused = one multi-tensor kernel per group instead of a Python loop over ~250 parameters
    fused = all(p.is_cuda for g in groups for p in g["params"])
    if os.environ.get("SNIPPER_OPT_PLAIN"):               # single-tensor reference implementation (debugging only)
        return torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4, foreach=False, fused=False)
    return torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4, capturable=capturable, fused=fused)


def _ranges(cpus):
    """[0,1,2,3,8,9] -> "0-3,8-9"."""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(index, sysfs="/sys"):
    """NUMA node of GPU ``index`` (torch's device order) from its PCI address, or None when the platform does not say
    (no sysfs entry, node -1, or a torch without the PCI fields)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"{sysfs}/bus/pci/devices/{addr}/numa_node").read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def cpu_busy(sample_s=0.15, proc_stat="/proc/stat"):
    """{cpu: fraction of the last ``sample_s`` seconds it was not idle} from two readings of /proc/stat, or {}."""
    def read():
        out = {}
        for line in open(proc_stat):
            f = line.split()
            if f and f[0].startswith("cpu") and f[0][3:].isdigit():
                v = [int(x) for x in f[1:]]
                out[int(f[0][3:])] = (sum(v), v[3] + (v[4] if len(v) > 4 else 0))      # total, idle + iowait
        return out
    try:
        a = read()
        time.sleep(sample_s)
        b = read()
        return {c: 1.0 - (b[c][1] - a[c][1]) / max(b[c][0] - a[c][0], 1) for c in b if c in a}
    except Exception:
        return {}


def _smt_sibling(cpu, sysfs="/sys"):
    try:
        sib = _parse_cpulist(open(f"{sysfs}/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read())
        return [c for c in sib if c != cpu]
    except Exception:
        return []


def pick_cpus(local_rank, n, allowed, node_of=gpu_numa_node, n_gpus=None, sysfs="/sys", busy=None):
    """The block of ``n`` CPUs this rank's two issuing threads are pinned to -> (cpus, numa node or None, rule).
    Rule "numa": CPUs of the NUMA node the rank's GPU hangs off (the reference starts 8 processes on a 2-socket host,
    README.md:67; a step here is within 20 % of its host-issue floor, so submitting to a GPU across the socket link is
    what would eat the scaling target), the k-th block of that node for the k-th GPU of that node.  Rule "block": the
    round-4 rule (block = local rank over the allowed CPUs) when the platform does not expose the topology.
    ``busy`` ({cpu: busy fraction}, cpu_busy()): rule "numa+idle" -- the node's CPUs are split into one share per GPU of the
    node and the rank takes the block of its share that was the least busy (a core's SMT sibling counted half) instead of
    the share's first block: the boxes are shared, and a step is host-bound as soon as a neighbour's thread sits on one of
    the two cores that issue it (24.6 instead of 21.3 ms measured on a box with a load average of 100)."""
    node = node_of(local_rank)
    if node is not None:
        try:
            node_cpus = [c for c in _parse_cpulist(open(f"{sysfs}/devices/system/node/node{node}/cpulist").read()) if c in set(allowed)]
            n_gpus = torch.cuda.device_count() if n_gpus is None else n_gpus
            k = sum(1 for j in range(min(local_rank, n_gpus)) if node_of(j) == node)      # GPUs of this node before mine
            if len(node_cpus) >= n:
                on_node = max(1, sum(1 for j in range(n_gpus) if node_of(j) == node))
                share = len(node_cpus) // on_node
                if busy and share >= n and k < on_node:
                    mine = node_cpus[k * share:(k + 1) * share]
                    cost = {c: busy.get(c, 0.0) + 0.5 * sum(busy.get(sb, 0.0) for sb in _smt_sibling(c, sysfs)) for c in mine}
                    best = min(range(0, len(mine) - n + 1), key=lambda st: (round(sum(cost[c] for c in mine[st:st + n]), 2), st))
                    return mine[best:best + n], node, "numa+idle"
                start = (k * n) % (len(node_cpus) - n + 1)
                return node_cpus[start:start + n], node, "numa"
        except Exception:
            pass
    start = (local_rank * n) % (len(allowed) - n + 1)
    return allowed[start:start + n], node, "block"


def msda_alg_bytes(d, bwd):
    """Algorithmic HBM bytes of one core-op launch; coordinates/weights are fp32."""
    e = d["esize"]
    re = d.get("row_esize", e)         # out / grad_out rows (bf16 under autocast: the kernels convert in place)
    rows = d["N"] * d["Lq"] * d["M"]
    v = d["N"] * d["S"] * d["M"] * d["D"]
    o = rows * d["D"]
    lp = rows * d["L"] * d["P"]
    ce = 4 if e == 2 else e
    if not bwd:
        return e * v + re * o + ce * 3 * lp
    ge = 4 if e == 2 else e            # grad_value accumulates in f32 for bf16 inputs
    return e * v + re * o + ge * v + ce * 6 * lp


PMC_PROFILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_pmc_bench_step.csv")


BWD_KERNELS = ("msda_bwd_d48_patchbin", "msda_bwd_d48_tile3", "msda_bwd_d48_far")   # query side; grad_value side (bf16 rows: two kernels, by tile size); far list


def pmc_traffic(path=PMC_PROFILE):
    """HBM-side bytes per launch of the owner-computes backward's two kernels from the committed rocprofv3 --pmc profile
    of this same command (FETCH_SIZE and WRITE_SIZE in separate passes, tools/collect_profiles.sh).  Returns (raw,
    corrected, why_not): raw = the counters as they read; corrected = with the gfx950 rule of MI355X_MICROARCH.md "HBM"
    applied -- FETCH_SIZE reports half the bytes of a 16-B-per-lane coalesced stream, which is how the grad_value-side
    kernel reads its grad_out rows and its tile, so that kernel's FETCH is doubled (WRITE_SIZE and the query-side kernel's
    12-B-per-lane gathers are left as read).  The profile's first line records the SOURCE HASH of the library that produced
    it (snipper_amd/build.py::source_hash); when it is not the hash of the library loaded now, the counters describe other
    kernels and (None, None, reason) is returned instead of stale numbers."""
    try:
        from snipper_amd import build as _build
        vals, prof_hash = {}, None
        for line in open(path):
            if line.startswith("# srchash="):
                prof_hash = line.split("=", 1)[1].strip()
                continue
            if line.startswith("#") or line.startswith("kernel,"):
                continue
            name, counter, _, kb = line.rsplit(",", 3)
            for tag in BWD_KERNELS:
                if tag in name:             # (per launch of the C call: the grad_value side's kernels add up)
                    vals[(tag, counter.strip())] = vals.get((tag, counter.strip()), 0.0) + float(kb) * 1024.0
        try:
            lib_hash = open(_build.HASH_PATH).read().strip()
        except OSError:
            lib_hash = None
        if prof_hash is None or lib_hash is None or prof_hash != lib_hash:
            return None, None, (f"profiles/{os.path.basename(path)} was collected with library source hash "
                                f"{(prof_hash or 'unrecorded')[:12]}, the loaded library is {(lib_hash or 'unrecorded')[:12]}: "
                                "counters not quoted (re-run tools/collect_profiles.sh)")
        if any((tag, c) not in vals for tag in BWD_KERNELS[:2] for c in ("FETCH_SIZE", "WRITE_SIZE")):
            return None, None, "the profile lacks FETCH_SIZE / WRITE_SIZE of the backward kernels"
        raw = sum(vals.values())
        corrected = raw + vals[(BWD_KERNELS[1], "FETCH_SIZE")]
        return int(raw), int(corrected), None
    except (OSError, ValueError) as e:
        return None, None, f"no PMC profile ({type(e).__name__})"


def kernel_census(run_step, n_steps: int = 2):
    """{"launches_per_step", "kernel_ms_per_step", "kernels_under_20us": {"launches_per_step", "ms_per_step"}, "memops_per_step"}
    from torch.profiler's device records of ``n_steps`` calls of ``run_step(i)`` (one unrecorded call first: the profiler's
    start-up allocations), or {"error": ...} when the platform's tracer yields no device records."""
    try:
        from torch.profiler import profile, ProfilerActivity
        run_step(0)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(n_steps):
                run_step(i)
            torch.cuda.synchronize()
        kernels, memops = [], 0
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA"):
                name = ev.name or ""
                dur = float(getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0) or 0.0)    # us
                if name.startswith(("Memcpy", "Memset")) or "memcpy" in name.lower() and "kernel" not in name.lower():
                    memops += 1
                else:
                    kernels.append(dur)
        if not kernels:
            return {"error": "the profiler returned no device kernel records"}
        small = [d for d in kernels if d < 20.0]
        return {"launches_per_step": round(len(kernels) / n_steps, 1),
                "kernel_ms_per_step": round(sum(kernels) / n_steps / 1e3, 3),
                "kernels_under_20us": {"launches_per_step": round(len(small) / n_steps, 1),
                                       "ms_per_step": round(sum(small) / n_steps / 1e3, 3)},
                "memops_per_step": round(memops / n_steps, 1),
                "source": f"torch.profiler device activity over {n_steps} eager steps after the timed region"}
    except Exception as e:                       # a measurement aid must never take the bench line down
        return {"error": f"{type(e).__name__}: {e}"}


def physical_cores():
    """(physical cores, the lscpu lines that say so) of the host."""
    import subprocess
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        get = lambda k: next((l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith(k)), None)
        cps, sockets, model = get("Core(s) per socket"), get("Socket(s)"), get("Model name")
        n = int(cps) * int(sockets)
        return n, f"lscpu: {model}; {sockets} socket(s) x {cps} cores, {get('Thread(s) per core')} thread(s) per core"
    except Exception:
        n = os.cpu_count() or 1
        return n, f"lscpu unavailable; os.cpu_count() = {n}"


def cpu_baseline(a, budget_s=45.0):
    """the CPU restatement of the reference's ``use_pytorch_deform=1`` path (oracle/, pinned to the
    reference's own outputs) timed on this box's host cores -- threads = physical cores -- best of a few repetitions after
    one warm-up, for (1) one encoder core call, (2) one encoder MSDeformAttn module forward + backward (1 snippet, T
    frames), (3) BASELINE configs[0]: T=1, enc2/dec4 transformer forward at the 600x800 geometry.  ``value`` is (2), the
    CPU counterpart of the metric's "MSDeformAttn fwd+bwd ms"; nothing is extrapolated to a whole training step."""
    from oracle import msda_oracle as O
    from snipper_amd.deformable_transformer import DeformableTransformer
    shapes = [(-(-a.height // s), -(-a.width // s)) for s in (8, 16, 32)]
    S = sum(h * w for h, w in shapes)
    C, M, L, P, T = a.hidden_dim, 8, 3, 4, a.frames
    cores, lscpu = physical_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    t_start = time.perf_counter()

    def best_of(fn, reps, share):
        fn()                                                   # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s * share:
                break
        return min(ts), len(ts)

    # (1) one encoder core call (N = 1): core_pytorch formulation
    sh = torch.tensor(shapes)
    value = r(1, S, M, C // M)
    loc = torch.rand(1, S, M, L, P, 2, generator=g)
    attn = torch.softmax(r(1, S, M, L * P), -1).view(1, S, M, L, P)
    with torch.no_grad():
        t_core, n_core = best_of(lambda: O.core_gridsample(value, sh, loc, attn), 5, 0.2)
    # (2) one encoder module forward + backward (per-pair spatiotemporal formulation, explicit weights)
    q = r(1, T, S, C).requires_grad_(True)
    src = r(1, T, S, C).requires_grad_(True)
    ref = torch.rand(1, T, S, L, 2, generator=g)
    vw, vb, ow, ob = r(C, C) * 0.05, torch.zeros(C), r(C, C) * 0.05, torch.zeros(C)
    offw, offb = [r(M * L * P * 2, C) * 0.01] * T, [r(M * L * P * 2)] * T
    attw, attb = [r(M * L * P, C) * 0.05] * T, [torch.zeros(M * L * P)] * T

    def module_step():
        q.grad = src.grad = None
        out, _, _ = O.st_msdeform_attn(q, ref, src, shapes, None, vw, vb, offw, offb, attw, attb, ow, ob, M, L, P, T)
        out.sum().backward()
    t_mod, n_mod = best_of(module_step, 3, 0.65)
    # (3) BASELINE configs[0]: T=1, enc2/dec4, hidden 384, 60 queries, transformer forward on 600x800 feature maps
    torch.manual_seed(0)
    tr = DeformableTransformer(d_model=C, nhead=M, num_encoder_layers=2, num_decoder_layers=4, dim_feedforward=1024,
                               dropout=0.1, activation="relu", return_intermediate_dec=True, num_feature_levels=L,
                               dec_n_points=P, enc_n_points=P, n_frame=1, n_future_frame=0, use_pytroch_deform=True,
                               num_keypoints=15).eval()
    srcs = [r(1, C, 1, h, w) for h, w in shapes]
    masks = [torch.zeros(1, C, 1, h, w, dtype=torch.bool) for h, w in shapes]
    pos = [r(1, C, 1, h, w) for h, w in shapes]
    qe = r(60, 2 * C)
    with torch.no_grad():
        t_cfg1, n_cfg1 = best_of(lambda: tr(srcs, masks, pos, qe), 2, 1.0)
    return {"value": round(t_mod * 1e3, 1), "unit": f"ms per encoder MSDeformAttn module fwd+bwd (1 snippet, T={T}, 600x800)",
            "cores": cores, "kind": "port",
            "sample": (f"oracle/ (use_pytorch_deform=1 formulation, pinned to the reference's outputs) on {cores} host threads "
                       f"= physical cores [{lscpu}]; best of {n_mod} after 1 warm-up; no extrapolation"),
            "timings_s": {"encoder_core_call_N1": round(t_core, 4), "encoder_module_fwd_bwd": round(t_mod, 4),
                          "config1_T1_enc2_dec4_transformer_forward": round(t_cfg1, 4)},
            "repetitions": {"core": n_core, "module": n_mod, "config1": n_cfg1},
            "total_cpu_seconds": round(time.perf_counter() - t_start, 1)}


def reference_module(mod, query, ref, src, hw, groups):
    """The REFERENCE's formulation of the MSDeformAttn module (models/ops/modules/ms_deform_attn.py:99-243 with
    ``use_pytorch_deform=1``) in plain PyTorch on the module's parameters: one value projection, then per (query frame,
    value frame) pair its own offset / logit Linears, a joint softmax over levels x points x neighbour frames, one
    grid_sample core op per pair, the sum of the pairs, the output projection.  nn.Linear everywhere (F.linear), float32:
    the denominator of BASELINE.json's ">= 3x the PyTorch-reference MSDeformAttn throughput"."""
    from snipper_amd.ms_deform_attn_func import ms_deform_attn_core_pytorch
    N, T1, Lq, C = query.shape
    T2, S = src.shape[1], src.shape[2]
    M, L, P = mod.n_heads, mod.n_levels, mod.n_points
    value = F.linear(src, mod.value_proj.weight, mod.value_proj.bias).view(N, T2, S, M, C // M)
    scale = torch.tensor([[w, h] for h, w in hw], dtype=query.dtype, device=query.device)
    outs = []
    for t1, grp in enumerate(groups):
        q = query[:, t1]
        logits = torch.stack([F.linear(q, mod.attention_weights[t2].weight, mod.attention_weights[t2].bias)
                              .view(N, Lq, M, L, P) for t2 in grp], -1)
        prob = F.softmax(logits.flatten(-3), -1).view(N, Lq, M, L, P, len(grp))
        acc = None
        for k, t2 in enumerate(grp):
            off = F.linear(q, mod.sampling_offsets[t2].weight, mod.sampling_offsets[t2].bias).view(N, Lq, M, L, P, 2)
            loc = ref[:, t1, :, None, :, None, :] + off / scale[None, None, None, :, None, :]
            o = ms_deform_attn_core_pytorch(value[:, t2], hw, loc, prob[..., k])
            acc = o if acc is None else acc + o
        outs.append(acc)
    return F.linear(torch.stack(outs, 1), mod.output_proj.weight, mod.output_proj.bias)


def time_msda_modules(a, device):
    """MSDeformAttn module fwd+bwd ms at the bench geometry (encoder Lq=S, decoder Lq=60), three ways on the same GPU:

      *_module_fwd_bwd_ms                       this package's HIP path in the step's precision;
      *_module_reference_fp32_fwd_bwd_ms        the REFERENCE's formulation (``reference_module`` above: per-pair
                                                grid_sample core op, per-pair Linears, joint softmax, F.linear
                                                everywhere, float32 -- the reference has no AMP): the denominator of
                                                BASELINE.json's ">= 3x the PyTorch-reference MSDeformAttn throughput";
      *_module_package_pytorch_switch_fwd_bwd_ms  this package's own module with ``use_pytorch_deform=1`` under the step's
                                                autocast: the tied single-grid_sample formulation plus this package's GEMM
                                                kernels -- NOT the reference's cost, reported for continuity with round 1.
    """
    from snipper_amd.ms_deform_attn import MSDeformAttn, frame_neighbours
    shapes = [(-(-a.height // s), -(-a.width // s)) for s in (8, 16, 32)]
    S = sum(h * w for h, w in shapes)
    sh = torch.tensor(shapes, device=device)
    sh._snipper_host = shapes
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    res = {}
    amp = a.precision == "bf16"
    for mode, Lq, T1 in (("encoder", S, a.frames), ("decoder", 60, a.frames + a.future_frames)):
        for kind in ("hip", "reference_fp32", "package_pytorch_switch"):
            mod = MSDeformAttn(a.hidden_dim, 3, 8, 4, a.frames, mode, kind == "package_pytorch_switch",
                               mode == "decoder").to(device)
            q = torch.randn(a.batch, T1, Lq, a.hidden_dim, device=device, requires_grad=True)
            src = torch.randn(a.batch, a.frames, S, a.hidden_dim, device=device, requires_grad=True)
            if mode == "encoder":   # the encoder's reference points are the pixel centres of the maps
                from snipper_amd.deformable_transformer import DeformableTransformerEncoder
                vr = torch.ones(a.batch, 3, 2, device=device)
                ref = DeformableTransformerEncoder.get_reference_points(sh, vr, device)[:, None].expand(-1, T1, -1, -1, -1)
            else:
                ref = torch.rand(a.batch, T1, Lq, 3, 2, device=device)
            groups = [frame_neighbours(t1, a.frames, a.frames) for t1 in range(T1)]

            def run():
                if kind == "reference_fp32":
                    o = reference_module(mod, q, ref, src, shapes, groups)
                else:
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):      # the step's precision
                        o = mod(q, ref, src, sh, lsi, None)
                o = o[0] if isinstance(o, tuple) else o
                o.float().sum().backward()
            key = f"{mode}_module_fwd_bwd_ms" if kind == "hip" else f"{mode}_module_{kind}_fwd_bwd_ms"
            try:
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 5
                for _ in range(n):
                    run()
                torch.cuda.synchronize()
                res[key] = round((time.perf_counter() - t0) / n * 1e3, 3)
            except RuntimeError as e:              # (the grid_sample formulation needs several GB at the encoder shape)
                if kind == "hip":
                    raise
                res[key.replace("_fwd_bwd_ms", "_error")] = str(e)[:80]
            del mod, q, src
            torch.cuda.empty_cache()
    for mode in ("encoder", "decoder"):
        a_, b_ = res.get(f"{mode}_module_fwd_bwd_ms"), res.get(f"{mode}_module_reference_fp32_fwd_bwd_ms")
        if a_ and b_:
            res[f"{mode}_speedup_vs_pytorch_reference"] = round(b_ / a_, 2)
        c_ = res.get(f"{mode}_module_package_pytorch_switch_fwd_bwd_ms")
        if a_ and c_:
            res[f"{mode}_speedup_vs_package_pytorch_switch"] = round(c_ / a_, 2)
    return res


def self_launch_command(argv, gpus: int, port: int = None, script: str = None):
    """The command line that starts ``gpus`` ranks of this script on one node (reference README.md:67 starts its 8 processes
    with one `torch.distributed.launch --nproc_per_node=8` command): one process per GPU under torch.distributed.run,
    rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    port = port or int(os.environ.get("MASTER_PORT", 0)) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(argv=None, script: str = None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process (nothing in
    this process has touched the GPU yet -- an exec from a process that has is forbidden on this pool), relay rank 0's JSON
    line on stdout, everything else on stderr, and return the child's exit code.  Returns None when there is nothing to
    launch (N = 1, or WORLD_SIZE is set: we ARE a rank)."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    if "WORLD_SIZE" in os.environ:
        return None
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1:
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = self_launch_command(argv, known.gpus, script=script)
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
                last_json = t
                continue
            except ValueError:
                pass
        sys.stderr.write(ln)
    rc = p.wait()
    if last_json is not None:
        sys.stdout.write(last_json + "\n")
        sys.stdout.flush()
    return rc


def set_offset_sigma(model, sigma_px: float, net, batch, amp: bool) -> None:
    """Give the encoder's sampling offsets a per-query spread of about ``sigma_px`` pixels (standard deviation of the
    weight-dependent part; the reference's 8-direction bias grid stays): ``sampling_offsets.weight`` is zero at
    initialisation (models/ops/modules/ms_deform_attn.py:82-90), so a freshly built model samples every query at the same
    <= 4 px bias grid -- the best case for the owner-computes backward, whose speed depends on how many taps lie within
    ``near_radius`` of the query's anchor.  A trained model spreads them; this emulates that.  The weight is drawn N(0, 1)
    and scaled per layer by sigma / mean ||query||_2, measured with one forward pass."""
    layers = list(model.transformer.encoder.layers)
    norms, hooks = {}, []
    for i, layer in enumerate(layers):
        def pre(mod, args, i=i):
            norms[i] = float(args[0].detach().float().norm(dim=-1).mean())
        hooks.append(layer.self_attn.register_forward_pre_hook(pre))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        net(list(batch[0]))
    for h in hooks:
        h.remove()
    gen = torch.Generator(device="cpu").manual_seed(1234)
    with torch.no_grad():
        for i, layer in enumerate(layers):
            w = layer.self_attn.sampling_offsets[0].weight
            w.copy_((torch.randn(w.shape, generator=gen) * (sigma_px / max(norms.get(i, 1.0), 1e-6))).to(w.device))
            # the scale was measured on THIS rank's batch: every replica takes rank 0's weights
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.broadcast(w.detach(), 0)


def main():
    rc = maybe_self_launch()
    if rc is not None:
        sys.exit(rc)
    # Native libraries write to the process's stdout too (RCCL prints a version banner when its first communicator comes
    # up): keep the real stdout for the ONE JSON line and send everything else that goes to fd 1 to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="snippets per GPU")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--future-frames", type=int, default=0)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--hidden-dim", type=int, default=384)
    ap.add_argument("--enc-layers", type=int, default=6)
    ap.add_argument("--dec-layers", type=int, default=6)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="bf16",
                    help="bf16 = autocast for the dense layers (BASELINE configs[2]); fp32 master weights, fp32 sampling")
    ap.add_argument("--use-pytorch-deform", type=int, default=0, help="1 = reference debug path (comparison only)")
    ap.add_argument("--loss", choices=["criterion", "surrogate"], default="criterion",
                    help="criterion = SetCriterion + Hungarian matcher as the reference trains; surrogate = fixed assignment")
    ap.add_argument("--master-weights", type=int, default=0,
                    help="bf16 only: 1 = bf16 parameters + fp32 master copy in the optimizer; 0 = fp32 parameters under autocast")
    ap.add_argument("--graph", type=int, default=0, help="1 = capture the step in a hipGraph (measured: no gain on ROCm 7.2, 73 vs 71 ms; off by default)")
    ap.add_argument("--ddp", choices=["flat", "torch"], default="flat",
                    help="N>1 gradient averaging: flat = one flat buffer + a few large all-reduces after backward "
                         "(snipper_amd/grad_sync.py); torch = DistributedDataParallel (host-bound: +15 ms/step)")
    ap.add_argument("--flat-params", type=int, default=1,
                    help="1 = one flat tensor per optimizer group (snipper_amd/flat_params.py): the same AdamW + clipping on "
                         "3 tensors instead of ~330; 0 = per-parameter form")
    ap.add_argument("--optimizer", choices=["flat-kernel", "torch"], default="flat-kernel",
                    help="with --flat-params 1: flat-kernel = clipping + AdamW as two launches of csrc/adamw_flat.cuh "
                         "(snipper_amd.flat_params.FlatAdamW, the arithmetic of torch.optim.AdamW); torch = clip_grad_norm_ + "
                         "torch.optim.AdamW (fused) on the three flat leaves")
    ap.add_argument("--pin-cores", type=int, default=8,
                    help="keep this process on a block of N neighbouring CPUs (block index = local rank); 0 = leave the "
                         "affinity alone")
    ap.add_argument("--gc-every", type=int, default=10,
                    help="collect garbage by hand every N steps and keep the automatic collector off in between (its "
                         "generation-0/1 passes cost the issuing thread ~1 ms per step); 0 = leave the collector alone")
    ap.add_argument("--offset-sigma-px", type=float, default=0.0,
                    help="spread of the encoder's sampling offsets in pixels for the TIMED region (0 = as initialised: every "
                         "query samples its <= 4 px bias grid); the default run also reports steps at 3 and 8 px in `locality`")
    ap.add_argument("--tile-kernel", type=int, default=0,
                    help="A/B aid: grad_value side of the encoder's owner-computes backward: 0 = library default (matrix pipe for "
                         "bf16 rows), 1 = the vector / LDS kernel of rounds 1-3")
    ap.add_argument("--no-locality-sweep", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline/msda extras (profiling runs)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:       # under a launcher the launcher's world size is authoritative
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # SNIPPER_SHARE_GPU=1 (path test only, never a measurement): all ranks of the job share the GPUs the box has -- with
    # SNIPPER_DIST_BACKEND=gloo two ranks can walk the whole N > 1 path (staged all-reduce hooks, rank barriers, gathers,
    # the cross-rank parameter check) on a 1-GPU box, where RCCL refuses two ranks on one device
    share_gpu = os.environ.get("SNIPPER_SHARE_GPU") == "1"
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if share_gpu else local_rank)
    torch.cuda.set_device(device)
    use_ddp = world > 1 or os.environ.get("SNIPPER_FORCE_DDP") == "1"   # the latter: 1-GPU test of the RCCL path
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("SNIPPER_DIST_BACKEND", "nccl")                         # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # The step is issued by two busy threads (Python + the autograd engine).  Left to the scheduler they wander over the
    # 256 logical CPUs of the box: measured host issue time 28.8-35.5 ms per step from run to run; on a block of
    # neighbouring cores 28.3-28.6 ms every time (12 alternating runs).  Each rank takes its own block of the CPUs it is
    # allowed to use; the affinity is restored before the CPU baseline.
    affinity0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pin_info = {"cpus": None, "numa_node": None, "rule": "unpinned"}
    if affinity0 is not None and a.pin_cores > 0 and len(affinity0) >= 2 * a.pin_cores:
        cpus, node, rule = pick_cpus(local_rank, a.pin_cores, sorted(affinity0),
                                     busy=cpu_busy() if os.environ.get("SNIPPER_PIN_IDLE", "1") != "0" else None)
        os.sched_setaffinity(0, set(cpus))
        pin_info = {"cpus": _ranges(cpus), "numa_node": node, "rule": rule}
        try:
            pin_info["loadavg"] = open("/proc/loadavg").read().split()[0]
        except OSError:
            pass
    from snipper_amd import MultiScaleDeformableAttention as MSDA
    from snipper_amd import _lib
    from snipper_amd.model import build_model
    _lib.load()                                                   # fail loudly if the HIP library is missing
    if a.tile_kernel:
        _lib.set_param("tile_kernel", a.tile_kernel)

    if os.environ.get("SNIPPER_BLAS"):                            # "cublas" = rocBLAS, "cublaslt" = hipBLASLt (aid)
        torch.backends.cuda.preferred_blas_library(os.environ["SNIPPER_BLAS"])
    torch.manual_seed(42 + rank)                                  # main.py:48,175
    margs = model_args(a)
    model = build_model(margs).to(device)
    # NHWC convolutions hand back NHWC weight gradients: keep the weights in the same format, so that DDP's
    # gradient buckets alias them without a strided copy
    model = model.to(memory_format=torch.channels_last)
    model.train()
    amp = a.precision == "bf16"
    masters = None
    if amp and a.master_weights:
        # Mixed precision with fp32 MASTER weights: the model's parameters are stored in bf16 (no per-call
        # autocast casts: ~780 tiny launches per step), the optimizer owns an fp32 copy, gradients are widened
        # and the updated masters narrowed back with multi-tensor copies.  FrozenBN statistics stay fp32.
        from snipper_amd.backbone import FrozenBatchNorm2d
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        master_named = [(n, p.detach().clone().float().requires_grad_(True)) for n, p in named]
        bn_names = ("weight", "bias", "running_mean", "running_var")
        bn_saved = [(mod, {k: getattr(mod, k).detach().clone() for k in bn_names})
                    for mod in model.modules() if isinstance(mod, FrozenBatchNorm2d)]
        model.to(torch.bfloat16)
        for mod, sv in bn_saved:
            for k, v in sv.items():
                setattr(mod, k, v)
        for (n, mp), (_, p) in zip(master_named, named):
            mp.grad = torch.zeros_like(mp)
        masters = ([mp for _, mp in master_named], [p for _, p in named])
        opt = build_optimizer(master_named, capturable=bool(a.graph))
    net, gsync, flatp, own_opt = model, None, None, None
    use_flat = bool(a.flat_params) and masters is None and not a.graph and not (use_ddp and a.ddp == "torch")
    if use_ddp and a.ddp == "torch":
        net = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[local_rank], broadcast_buffers=False, gradient_as_bucket_view=True,
            bucket_cap_mb=50, static_graph=True)
    elif use_ddp:
        # the same semantics (rank 0's initial weights everywhere; mean of the ranks' gradients after backward)
        # without a reducer hook per parameter: see snipper_amd/grad_sync.py
        from snipper_amd.grad_sync import FlatGradSync
        g_main, g_backbone, g_slow = optimizer_groups(list(model.named_parameters()))
        # flat layout [main | slow | backbone]: group by group so that the optimizer's flat leaves (flat_params.py) line up
        # with it, and in stages that backward completes one after the other (grad_sync_stages)
        stage_names = []
        gsync = FlatGradSync(g_main + g_slow + g_backbone, stages=grad_sync_stages(model, g_main + g_slow, stage_names))
        gsync.broadcast_parameters(list(model.parameters()) + list(model.buffers()))
    if masters is None:
        if use_flat:
            from snipper_amd.flat_params import FlatParameters
            g_main, g_backbone, g_slow = optimizer_groups(list(model.named_parameters()))
            if gsync is not None:
                assert [id(p) for p in gsync.params] == [id(p) for p in g_main + g_slow + g_backbone]
            flatp = FlatParameters([g_main, g_slow, g_backbone], grad_flat=gsync.flat if gsync is not None else None,
                                   grad_guard=gsync.slice_is_free if gsync is not None else None)
        opt = build_optimizer(list(model.named_parameters()), capturable=bool(a.graph), flat=flatp)
        if flatp is not None and a.optimizer == "flat-kernel" and not a.graph and not os.environ.get("SNIPPER_OPT_PLAIN"):
            from snipper_amd.flat_params import FlatAdamW
            # groups in FlatParameters' order (main, slow, backbone): the reference's learning rates (main.py:201-221)
            own_opt = FlatAdamW(flatp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4,
                                reference_groups=reference_param_groups(list(model.named_parameters())))
    batches = make_batches(a, device, 2, seed=1000 + rank)

    criterion = None
    if a.loss == "criterion":
        from snipper_amd.criterion import build_criterion
        criterion = build_criterion(criterion_args(a)).to(device)

    clip_params = [p for p in model.parameters() if p.requires_grad]   # walked once, not once per step

    host_ms = {}                          # SNIPPER_HOST_REGIONS: host time per region of the step (development aid)
    region_on = [False]

    def mark(name, t0):
        if region_on[0]:
            host_ms[name] = host_ms.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)
        return time.perf_counter()

    # SNIPPER_REGION_EVENTS=<steps>: at every region boundary of a step the host's clock AND an event on the stream -- where the
    # GPU reaches a boundary only just after the host issued it, the GPU is waiting for the host there (development aid)
    boundary_log = []
    boundary_on = [False]

    def boundary(name):
        if boundary_on[0]:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            boundary_log.append((name, time.perf_counter(), ev))

    if os.environ.get("SNIPPER_REGION_EVENTS"):
        def bounded(mod, name):
            def after(m, a, o):
                boundary("fwd." + name + " end")
                first = o
                while isinstance(first, (tuple, list, dict)):
                    first = next(iter(first.values())) if isinstance(first, dict) else first[0]
                first = getattr(first, "tensors", first)
                if torch.is_tensor(first) and first.requires_grad:
                    first.register_hook(lambda g: (boundary("bwd reaches " + name + " output"), None)[1])
            mod.register_forward_hook(after)
        bounded(model.backbone, "backbone")
        bounded(model.transformer.encoder, "encoder")
        bounded(model.transformer.decoder, "decoder")

    if os.environ.get("SNIPPER_HOST_REGIONS"):
        def timed(mod, name):
            mod.register_forward_pre_hook(lambda m, a: setattr(m, "_t0", time.perf_counter()))
            mod.register_forward_hook(lambda m, a, o: (mark(name, m._t0), None)[1])
        timed(model.backbone, "fwd.backbone")
        timed(model.transformer.encoder, "fwd.encoder")
        timed(model.transformer.decoder, "fwd.decoder")

    def train_step(imgs, tgt):
        t = time.perf_counter()
        boundary("step start")
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out, _ = net(list(imgs))
        t = mark("fwd.model(total)", t)
        boundary("fwd end")
        if criterion is not None:        # Hungarian matching + the six loss families (models/model.py:240-545)
            losses, _ = criterion(out, tgt["targets"])
            loss = criterion.weighted_sum(losses)
        else:
            loss = surrogate_loss(out, tgt)
        t = mark("criterion", t)
        boundary("criterion end")
        if masters is None:
            if flatp is not None:
                flatp.drop_param_grads()
            else:
                opt.zero_grad(set_to_none=True)
            loss.backward()
            t = mark("backward", t)
            boundary("backward end")
            if gsync is not None:
                gsync.sync()
            if own_opt is not None:
                flatp.pack()                 # (after sync() the gradients already live in the shared flat buffer)
                t = mark("clip", t)
                own_opt.step(0.1)            # global-norm clipping (engine.py:74) + AdamW in two launches; bumps the versions
            else:
                if flatp is not None:
                    flatp.pack()
                    torch.nn.utils.clip_grad_norm_(flatp.leaves, 0.1)
                else:
                    torch.nn.utils.clip_grad_norm_(clip_params, 0.1)
                t = mark("clip", t)
                opt.step()
                if flatp is not None:
                    flatp.after_step()
            t = mark("optimizer", t)
            boundary("optimizer end")
        else:
            mp, pp = masters
            for p in pp:
                p.grad = None
            loss.backward()
            with torch.no_grad():
                torch._foreach_copy_([m.grad for m in mp], [p.grad for p in pp])  # bf16 -> fp32, multi-tensor
            torch.nn.utils.clip_grad_norm_(mp, 0.1)
            opt.step()
            with torch.no_grad():
                torch._foreach_copy_(pp, mp)                                      # fp32 -> bf16, multi-tensor
        return loss

    def step(i):
        loss = train_step(*batches[i % len(batches)])
        ge = a.gc_every
        if ge and i % ge == ge - 1:
            import gc
            gc.collect(1)
        if os.environ.get("SNIPPER_PRINT_LOSS"):          # per-step loss (synchronises: debugging only)
            with torch.no_grad():
                prev = getattr(step, "prev", None)
                cur = [q.detach().clone() for q in clip_params]
                extra = ""
                if prev is not None:
                    d = [(c - q) for c, q in zip(cur, prev)]
                    extra = (f" |dp| {float(torch.stack([x.norm() for x in d]).norm()):.5f}"
                             f" max {max(float(x.abs().max()) for x in d):.6f}"
                             f" n_changed {sum(int((x != 0).any()) for x in d)}/{len(d)}")
                step.prev = cur
            print(f"[bench] step loss {float(loss.detach()):.4f}{extra}", file=sys.stderr)
        return loss

    def fence():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    if a.offset_sigma_px > 0:
        set_offset_sigma(model, a.offset_sigma_px, net, batches[0], amp)
    for i in range(a.warmup):
        step(i)
    fence()
    gc_every = a.gc_every
    if gc_every:
        # the collector's generation-0/1 passes run on the thread that issues the step (a few hundred autograd / tensor
        # objects per step trigger them constantly): collect by hand every N steps instead, as training loops commonly do
        import gc
        gc.collect()
        gc.disable()

    # ---- hipGraph: the step is ~3000 launches of mostly short kernels and the host cannot issue them as fast
    #      as the GPU retires them; all shapes are static, so the whole step (forward, loss, backward, RCCL
    #      all-reduce, clipping, AdamW) is captured once and replayed.  Inputs are copied into static buffers.
    graph, static_loss, graph_note = None, None, "eager"
    if a.graph and a.loss == "criterion":
        # capture was only ever exercised with the surrogate loss (73 vs 71 ms then); the criterion's target lists
        # are host-side Python and today's step crashes the capture, so the flag is honoured only with --loss surrogate
        print("[bench] --graph 1 needs --loss surrogate; running eagerly", file=sys.stderr)
        a.graph = 0
    if a.graph:
        try:
            static_imgs = torch.empty_like(batches[0][0])
            static_tgt = {k: torch.empty_like(v) for k, v in batches[0][1].items() if torch.is_tensor(v)}

            def load(i):
                imgs, tgt = batches[i % len(batches)]
                static_imgs.copy_(imgs)
                for k in static_tgt:
                    static_tgt[k].copy_(tgt[k])

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(3 if not use_ddp else 11):    # DDP wants > 10 eager iterations before capture
                    load(i)
                    train_step(static_imgs, static_tgt)
            torch.cuda.current_stream().wait_stream(side)
            fence()
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                static_loss = train_step(static_imgs, static_tgt)
            fence()

            def step(i):                                      # noqa: F811  (replaces the eager step)
                load(i)
                graph.replay()
                return static_loss
            for i in range(2):
                step(i)
            fence()
            graph_note = "hipGraph replay of the whole step"
        except Exception as e:                                # capture is an optimisation, never a requirement
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

            def step(i):                                      # noqa: F811
                return train_step(*batches[i % len(batches)])

    if os.environ.get("SNIPPER_CPROFILE"):        # host-side profile of the step (development aid)
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ.get("SNIPPER_CPROFILE_SORT", "tottime")).print_stats(int(os.environ.get("SNIPPER_CPROFILE_N", "45")))
    if os.environ.get("SNIPPER_TORCH_PROFILE"):   # op-level GPU time with input shapes (development aid)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=400,
                                                                   max_name_column_width=40, max_shapes_column_width=70),
              file=sys.stderr)
    if os.environ.get("SNIPPER_TORCH_PROFILE_CPU"):   # host time per operator, both threads (development aid)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=70, max_name_column_width=60),
              file=sys.stderr)
    if os.environ.get("SNIPPER_REGION_EVENTS"):
        n_ev = max(int(os.environ["SNIPPER_REGION_EVENTS"]), 4)
        torch.cuda.synchronize()
        boundary_on[0] = True
        for i in range(n_ev):
            step(i)
        torch.cuda.synchronize()
        boundary_on[0] = False
        # per boundary: median over the steps (the first two dropped) of host ms and GPU ms since the step's start, and of
        # the GPU's lag behind the host there (GPU clock aligned to the host's at the FIRST boundary of the run, where the GPU
        # was idle: lag = how long after the host issued the boundary the GPU reached it)
        name0, h0, e0 = boundary_log[0]
        steps_log, cur = [], []
        for name, h, e in boundary_log:
            if name == "step start" and cur:
                steps_log.append(cur); cur = []
            cur.append((name, 1e3 * (h - h0), e0.elapsed_time(e)))
        steps_log.append(cur)
        steps_log = [s_ for s_ in steps_log[2:] if len(s_) == len(steps_log[-1])]
        med = lambda v: sorted(v)[len(v) // 2]
        print("[bench] boundary                          host ms   GPU ms   GPU lag behind host (ms)", file=sys.stderr)
        rows_out = []
        for k, (name, _, _) in enumerate(steps_log[-1]):
            hs_ = med([s_[k][1] - s_[0][1] for s_ in steps_log])
            gs_ = med([s_[k][2] - s_[0][2] for s_ in steps_log])
            lag = med([s_[k][2] - s_[k][1] for s_ in steps_log])
            rows_out.append({"boundary": name, "host_ms": round(hs_, 3), "gpu_ms": round(gs_, 3), "gpu_lag_ms": round(lag, 3)})
            print(f"[bench] {name:36s} {hs_:8.3f} {gs_:8.3f} {lag:8.3f}", file=sys.stderr)
        print("[bench] region_events " + json.dumps(rows_out), file=sys.stderr)
    if os.environ.get("SNIPPER_ISSUE_TIME"):      # is the host or the GPU the limiter?  (development aid)
        # host time to ISSUE a few steps (no synchronisation) against the time until the GPU has retired them
        n_issue = int(os.environ.get("SNIPPER_ISSUE_TIME", "3")) if os.environ.get("SNIPPER_ISSUE_TIME", "1").isdigit() else 3
        n_issue = max(n_issue, 3)
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for i in range(n_issue):
            step(i)
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        h2 = time.perf_counter()
        print(f"[bench] issue {1e3 * (h1 - h0) / n_issue:.2f} ms/step, retire {1e3 * (h2 - h0) / n_issue:.2f} ms/step "
              f"({n_issue} steps)", file=sys.stderr)
    if os.environ.get("SNIPPER_CPROFILE"):        # where the host's issue time goes, by Python function (development aid)
        import cProfile, pstats
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for i in range(10):
            step(i)
        pr.disable()
        torch.cuda.synchronize()
        for key in ("tottime", "cumulative"):
            st = pstats.Stats(pr, stream=sys.stderr)
            st.sort_stats(key).print_stats(45)
    if os.environ.get("SNIPPER_HOST_REGIONS"):
        torch.cuda.synchronize()
        region_on[0] = True
        for i in range(10):
            step(i)
        region_on[0] = False
        torch.cuda.synchronize()
        print("[bench] host ms per step by region: " +
              ", ".join(f"{k} {v / 10:.2f}" for k, v in host_ms.items()), file=sys.stderr)
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if gsync is not None:
        gsync.check_errors()          # a late gradient flagged by the LAST step's sync() would otherwise never be reported
    # ---- how long the HOST needs to issue a step: three steps issued back to back on an idle GPU, the
    #      clock read BEFORE the device is waited for.  While this stays below ms_per_step the step is GPU-bound; a rank whose
    #      figure exceeds its ms_per_step is host-bound (a slow or shared core), which a per-rank step time alone cannot tell
    #      from a slow link.  Every rank measures (the steps contain the collectives), rank 0 reports.
    host_issue_ms = None
    if not a.no_extras and graph is None:
        rounds = []
        for r in range(3):         # best of three rounds (shared hosts: a neighbour's burst on an issuing core shows up in one of them)
            h0 = time.perf_counter()
            for i in range(3):
                step(a.warmup + a.steps + 3 * r + i)
            rounds.append((time.perf_counter() - h0) / 3 * 1e3)
            fence()
        host_issue_ms = round(min(rounds), 3)
    loss_val = float(loss.detach())
    per_rank_ms = None
    if use_ddp:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(x) / a.steps * 1e3, 3) for x in every]
        per_rank_pin = [None] * world               # every rank's CPU set and its GPU's NUMA node
        dist.all_gather_object(per_rank_pin, pin_info)
        per_rank_issue = [None] * world             # every rank's host-issue time per step
        dist.all_gather_object(per_rank_issue, host_issue_ms)
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # data-parallel invariant (DDP's, main.py:193-196): after any number of steps every rank holds the SAME parameters
        # -- same initial broadcast, same averaged gradients, same deterministic update.  Checked bit for bit through two
        # order-sensitive checksums of all trainable parameters (sum and index-weighted sum in float64).
        with torch.no_grad():
            ps = [q.detach().double().flatten() for q in model.parameters() if q.requires_grad]
            chk = torch.stack([torch.stack([q.sum() for q in ps]).sum(),
                               torch.stack([(q * torch.arange(1, q.numel() + 1, device=q.device, dtype=q.dtype)).sum()
                                            for q in ps]).sum()])
        every_chk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(every_chk, chk)
        params_in_sync = all(bool(torch.equal(c, every_chk[0])) for c in every_chk)
        if not params_in_sync:       # reported, not raised: the line (and the other ranks' collectives) must still complete
            print(f"[bench] rank {rank}: PARAMETERS DIFFER ACROSS RANKS after {a.warmup + a.steps} steps: "
                  f"{[c.tolist() for c in every_chk]}", file=sys.stderr)

    # kernel launch durations for the roofline: events on the launch stream around every core-op launch.  Under
    # graph replay there is no per-launch host hook, so the same step is run eagerly (same kernels, same
    # inputs) right after the timed region -- when the region itself was eager the events sit inside it.
    launches, dense_launches = [], []
    if (rank == 0) and not a.no_extras:
        from snipper_amd import dense as _dense
        MSDA.enable_launch_timing(True)
        _dense.enable_launch_timing(True)
        for i in range(2):
            train_step(*batches[i % len(batches)])
        launches = MSDA.launch_timings()
        dense_launches = _dense.launch_timings()
        MSDA.enable_launch_timing(False)
        _dense.enable_launch_timing(False)
        if os.environ.get("SNIPPER_DENSE_TABLE"):          # tools/dense_roofline.py: every GEMM-shaped launch of two steps
            with open(os.environ["SNIPPER_DENSE_TABLE"], "w") as fh:
                json.dump([[k, list(sh), f, b, ms] for k, sh, f, b, ms in dense_launches], fh)
    if use_ddp and not a.no_extras:      # the other ranks must run the same eager steps (collectives inside)
        if rank != 0:
            for i in range(2):
                train_step(*batches[i % len(batches)])
        fence()

    # ---- kernel census of a step: every device kernel of two eager steps from the profiler's device activity
    #      (roctracer): launches per step, their summed duration, and the part of it in kernels shorter than 20 us -- the
    #      launch-bound tail, whose kernels cannot fill 256 CUs.  Memcpy / memset records are counted apart.
    #      One GPU only: with N > 1 a profiler failure on rank 0 in the middle of a step would leave the other ranks waiting in
    #      that step's collectives, and the launch count does not depend on N.
    census = None
    if rank == 0 and world == 1 and not use_ddp and not a.no_extras and graph is None:
        census = kernel_census(lambda i: train_step(*batches[i % len(batches)]), 2)
    if os.environ.get("SNIPPER_ATEN_OPS") and rank == 0 and world == 1:      # development aid: the ATen residue of a step by operator
        from torch.profiler import profile, ProfilerActivity
        train_step(*batches[0])
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            train_step(*batches[0])
            torch.cuda.synchronize()
        rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and (getattr(e, "self_device_time_total", 0) or 0) > 0]
        rows.sort(key=lambda e: -e.self_device_time_total)
        print(f"[aten] {sum(e.self_device_time_total for e in rows):.0f} us of device time in {sum(e.count for e in rows)} operator calls",
              file=sys.stderr)
        for e in rows[:40]:
            print(f"[aten] {e.key:32s} calls {e.count:4d} device_us {e.self_device_time_total:9.1f}", file=sys.stderr)

    # ---- where the staged all-reduces are launched on the GPU timeline of a step (events on the compute stream): the stages
    #      launched from autograd hooks must sit INSIDE backward.  (On one GPU with a 1-rank group RCCL launches no kernel for
    #      an in-place all-reduce, so a kernel trace cannot show them; the launch points can.)
    sync_trace = None
    if gsync is not None and not a.no_extras and graph is None:
        imgs_, tgt_ = batches[0]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out_, _ = net(list(imgs_))
        if criterion is not None:
            l_, _ = criterion(out_, tgt_["targets"])
            loss_ = criterion.weighted_sum(l_)
        else:
            loss_ = surrogate_loss(out_, tgt_)
        if flatp is not None:
            flatp.drop_param_grads()
        else:
            opt.zero_grad(set_to_none=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gsync.trace = []
        e0.record()
        loss_.backward()
        e1.record()
        gsync.sync()
        torch.cuda.synchronize()
        bwd_ms = e0.elapsed_time(e1)
        at = {i: e0.elapsed_time(ev) for i, ev in gsync.trace}
        stage_mb = [round((st.hi_elem - st.lo_elem) * 4 / 1e6, 2) for st in gsync.stages]
        total_mb = gsync.flat.numel() * 4 / 1e6
        early_mb = sum(mb for i, mb in enumerate(stage_mb) if i in at and at[i] <= 0.9 * bwd_ms)
        # what enters the wire in the last 20 % of backward (or only from sync(), after it): the exposed tail at N > 1
        late20_mb = total_mb - sum(mb for i, mb in enumerate(stage_mb) if i in at and at[i] <= 0.8 * bwd_ms)
        sync_trace = {"backward_ms": round(bwd_ms, 3),
                      "stage_launch_ms_after_backward_start": [[i, round(at[i], 3)] for i in sorted(at)],
                      "stage_mbytes": stage_mb, "total_mbytes": round(total_mb, 2),
                      "fraction_of_bytes_launched_before_last_10pct_of_backward": round(early_mb / total_mb, 3),
                      "mbytes_launched_in_last_20pct_of_backward_or_later": round(late20_mb, 2),
                      "stages": stage_names[:len(gsync.stages)]}
        gsync.trace = None
        gsync.check_errors()
        if flatp is not None:
            flatp.pack()

    # ---- the same step at trained-like locality: sigma 3 px and 8 px, a few steps each, after the contract's timed region; one GPU only
    locality = []

    def owner_bwd_ms(ls):
        t = [ms for kind, variant, d, ms in ls if kind == "bwd" and d["Lq"] > 1000]
        return round(sum(t) / len(t), 4) if t else None

    if rank == 0 and world == 1 and not a.no_extras and not a.no_locality_sweep and graph is None:
        locality.append({"offset_sigma_px": a.offset_sigma_px, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
                         "encoder_bwd_ms_per_launch": owner_bwd_ms(launches), "note": "the timed region"})
        saved_offsets = [l.self_attn.sampling_offsets[0].weight.detach().clone() for l in model.transformer.encoder.layers]
        for sig in (3.0, 8.0):
            if sig == a.offset_sigma_px:
                continue
            set_offset_sigma(model, sig, net, batches[0], amp)
            for i in range(6):
                step(i)
            torch.cuda.synchronize()
            # per-step times from events on the compute stream, median of 16: a short window right after the weights were
            # rewritten is exposed to one-off host stalls (allocator growth, a collector pass), which the contract's
            # timed region is not; the mean is reported beside it
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
            t1 = time.perf_counter()
            for i in range(16):
                evs[i].record()
                step(i)
            evs[16].record()
            torch.cuda.synchronize()
            mean_ms = (time.perf_counter() - t1) / 16 * 1e3
            per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(16))
            ms = 0.5 * (per[7] + per[8])
            MSDA.enable_launch_timing(True)
            train_step(*batches[0])
            ls = MSDA.launch_timings()
            MSDA.enable_launch_timing(False)
            locality.append({"offset_sigma_px": sig, "ms_per_step": round(ms, 3), "encoder_bwd_ms_per_launch": owner_bwd_ms(ls),
                             "mean_ms_per_step": round(mean_ms, 3), "note": "median of 16 steps after 6 warm-up steps"})
        with torch.no_grad():                  # what runs after the sweep (module timings) sees the model as it was
            for l, w0 in zip(model.transformer.encoder.layers, saved_offsets):
                l.self_attn.sampling_offsets[0].weight.copy_(w0)
            torch.autograd.graph.increment_version([l.self_attn.sampling_offsets[0].weight for l in model.transformer.encoder.layers])

    # (only now: the extra steps above -- launch timings, gradient-sync trace, locality sweep -- run under the host conditions of
    #  the timed region; unpinned and with the automatic collector on, the sigma = 8 px sweep was host-bound in some runs: 30.1
    #  against 26.1 ms per step)
    if gc_every:
        import gc
        gc.enable()
    if affinity0 is not None:               # every thread: the intra-op pool's workers were born with the narrow mask
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), affinity0)
            except OSError:
                pass
    if rank == 0:
        snippets = a.batch * world * a.steps
        line = {
            "metric": "snippets/sec (T=4, 600x800, hidden_dim=384) training step; MSDeformAttn fwd+bwd ms",
            "value": round(snippets / elapsed, 4), "unit": "snippets/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not amp else "bf16", "data": "synthetic",
            "config": {"workload": (f"T={a.frames}+{a.future_frames} enc{a.enc_layers}/dec{a.dec_layers} "
                                    f"hidden_dim={a.hidden_dim} L=3 nq=60 {a.height}x{a.width} ResNet-50, "
                                    f"train step fwd+bwd+clip+AdamW (" +
                                    ("BASELINE configs[4], the forecast model, on one node's worth of ranks" if a.future_frames else
                                     "BASELINE configs[2]/[3]" if (a.height, a.width) == (600, 800) else
                                     "the reference README's JTA / Panoptic input size, README.md:145-146,208-209") + ")"),
                       "global_batch": a.batch * world, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                       "loss": ("SetCriterion + Hungarian matcher (reference coefficients, 6+8 synthetic persons)"
                                if a.loss == "criterion" else "fixed-assignment surrogate"),
                       "backbone_convs": ("every ResNet-50 convolution on own bf16 MFMA kernels (BN / residual / ReLU fused): 7x7 "
                                          "stem, 1x1 and 3x3 forward, data and weight gradients; nothing in MIOpen"),
                       "grad_sync": ("none (1 GPU)" if not use_ddp else
                                     ("DistributedDataParallel" if a.ddp == "torch" else
                                      "flat buffer, 8 stages (decoder side, upper encoder half, the rest, layer4 block by block, layer3, "
                                      "layer2) all-reduced over RCCL from autograd hooks while backward runs (snipper_amd/grad_sync.py)")),
                       "layer_calls": ("one native call and one autograd node per encoder layer and per decoder layer and direction "
                                       "(include/snipper_layers.h; the same launches as the per-module sequence, bit-identical)"
                                       if (os.environ.get("SNIPPER_ENC_NATIVE", "1") != "0" and os.environ.get("SNIPPER_DEC_NATIVE", "1") != "0"
                                           and amp) else "per-module autograd nodes (SNIPPER_ENC_NATIVE / SNIPPER_DEC_NATIVE = 0, or float32)"),
                       "msda_path": "pytorch grid_sample" if a.use_pytorch_deform else
                                    ("snipper_amd HIP (tied single-launch" +
                                     ("; encoder's bf16 temporal mean and grad_value head-major [n, head, position, 48] inside the "
                                      "module core, snipper_msda_config.value_layout = 1)"
                                      if amp and os.environ.get("SNIPPER_VALUE_HEAD_MAJOR", "1") != "0" else ")")),
                       "launch": graph_note,
                       "weights": ("bf16 parameters + fp32 master weights" if masters is not None else
                                   "fp32 parameters" + (" under bf16 autocast" if amp else "")),
                       "host": (f"gc.collect every {a.gc_every} steps, automatic collector off" if a.gc_every else "default gc") +
                               (f"; process pinned to {a.pin_cores} neighbouring CPUs: {pin_info['cpus']} "
                                f"(rule {pin_info['rule']}, GPU's NUMA node {pin_info['numa_node']}; host load average "
                                f"{pin_info.get('loadavg', '?')})" if a.pin_cores else ""),
                       "optimizer": ("global-norm clipping + AdamW on the flat parameter buffer in two launches "
                                     "(snipper_amd.flat_params.FlatAdamW, csrc/adamw_flat.cuh: the arithmetic of "
                                     "torch.optim.AdamW + clip_grad_norm_)" if own_opt is not None else
                                     "torch.optim.AdamW (fused) + clip_grad_norm_ on one flat tensor per group "
                                     "(snipper_amd/flat_params.py)" if flatp is not None else
                                     "torch.optim.AdamW (fused) + clip_grad_norm_ per parameter")},
            "final_loss": round(loss_val, 5),
        }
        if use_ddp:
            # what RCCL itself saw -- a SCALE record can be checked against it
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
            line["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                                   "per_rank_ms_per_step": per_rank_ms,
                                   "per_rank_ms_per_step_min_max": [min(per_rank_ms), max(per_rank_ms)],
                                   "per_rank_cpu_pinning": per_rank_pin,
                                   "per_rank_host_issue_ms_per_step": per_rank_issue,
                                   "parameters_identical_across_ranks": params_in_sync,
                                   "shared_gpu_path_test": share_gpu,
                                   "grad_allreduce_mbytes_per_step": round(gsync.flat.numel() * 4 / 1e6, 2) if gsync is not None else None}
        line["host_issue_ms"] = host_issue_ms          # host time to issue one step (no device wait); < ms_per_step = GPU-bound
        if census is not None:
            line["launches_per_step"] = census.get("launches_per_step")
            line["kernels_under_20us_ms"] = (census.get("kernels_under_20us") or {}).get("ms_per_step")
            line["kernel_census"] = census
        if locality:
            line["locality"] = locality
            # the representative operating point once training has left the initialisation: the same step with
            # the encoder's sampling offsets spread by N(0, 3 px) around the reference's bias grid
            s3 = [x for x in locality if x["offset_sigma_px"] == 3.0]
            if s3:
                line["ms_per_step_at_offset_sigma_3px"] = s3[0]["ms_per_step"]
                line["value_at_offset_sigma_3px"] = round(a.batch * world / (s3[0]["ms_per_step"] * 1e-3), 4)
        if sync_trace:
            line["grad_sync_trace"] = sync_trace
        if launches:
            by = {}
            for kind, variant, d, ms in launches:
                key = (kind, variant, d["N"], d["Lq"])
                by.setdefault(key, [d, []])[1].append(ms)
            tot = {k: sum(v[1]) for k, v in by.items()}
            dom = max(tot, key=tot.get)
            d, times = by[dom]
            avg_ms = sum(times) / len(times)
            bts = msda_alg_bytes(d, dom[0] == "bwd")
            ach = bts / (avg_ms * 1e-3) / 1e9
            raw, corrected, why_not = (pmc_traffic() if dom[1] == "d48_owner_mfma" and d["N"] == 8 and d["Lq"] == 9875 else
                                       (None, None, "the dominant launch is not the bf16 encoder backward the PMC profile covers"))
            line["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": corrected, "traffic_raw_counters": raw,
                                "traffic_source": ("NOT measured in this run: read from the committed profiles/r06_pmc_bench_step.csv "
                                                   "(collected with THIS library: its source hash is recorded in the file and checked) = "
                                                   "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, tools/collect_profiles.sh) of "
                                                   "this command at the same kernels, query-side + grad_value-side kernel; "
                                                   "`traffic` doubles the latter's FETCH_SIZE (gfx950 counts half of a "
                                                   "16-B-per-lane stream: MI355X_MICROARCH.md, HBM), `traffic_raw_counters` "
                                                   "is what the counters read" if corrected else why_not),
                                "kernel": f"msda_{dom[0]}_{dom[1]} N={d['N']} Lq={d['Lq']}",
                                "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": bts,
                                "launches_timed": len(times)}
            # the same launch at trained-like offsets ("put the sigma = 3 figure beside the sigma = 0 one"):
            # reference bias grid + N(0, sigma) px per coordinate, from the locality sweep after the timed region
            if dom[0] == "bwd" and locality:
                line["roofline"]["at_offset_sigma_px"] = {
                    str(x["offset_sigma_px"]): {"avg_launch_ms": x["encoder_bwd_ms_per_launch"],
                                                "frac": round(bts / (x["encoder_bwd_ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                    for x in locality if x.get("encoder_bwd_ms_per_launch")}
            line["msda_launch_ms_per_step"] = {f"{k[0]}_{k[1]}_N{k[2]}_Lq{k[3]}": round(v / 2, 3) for k, v in tot.items()}
        if dense_launches:
            # the dense parts against the MFMA peak.  Aggregate over every GEMM-shaped launch of this
            # library in two eager steps (linear / weight-stationary / NN data gradient / weight gradient / 3x3 / stem; the
            # decoder-size float32 products are not in it): FLOPs of the products as defined (no padding) over the sum of
            # the launches' durations (events on the launch stream); per-layer table: profiles/r06_backbone_roofline.csv
            fl = sum(x[2] for x in dense_launches)
            ms = sum(x[4] for x in dense_launches)
            by_kind = {}
            for k, sh, f, b, t in dense_launches:
                e = by_kind.setdefault(k, [0, 0.0, 0.0])
                e[0] += 1; e[1] += f; e[2] += t
            line["roofline_dense"] = {
                "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "launches_per_step": len(dense_launches) // 2,
                "ms_per_step": round(ms / 2, 3), "gflop_per_step": round(fl / 2 / 1e9, 1),
                "by_kernel": {k: {"launches_per_step": v[0] // 2, "ms_per_step": round(v[2] / 2, 3),
                                  "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 1)} for k, v in sorted(by_kind.items())}}
        if not a.no_extras:
            line["msda"] = time_msda_modules(a, device)
        if world == 1 and not a.no_cpu_baseline and not a.no_extras:
            line["cpu_baseline"] = cpu_baseline(a)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
