#!/usr/bin/env python3
"""One table of the numbers that matter, generated from a bench record.

    python tools/numbers_table.py BENCH_r05.json            # the driver's record ({"parsed": {...}} or the bare JSON line)
    python tools/numbers_table.py gpurun_out/r05_bench.log  # a log whose last JSON line is bench.py's

Prints a markdown table: workload, kernel / quantity, milliseconds, fraction of its roofline, what the fraction is of, and the
key of the record the number was read from -- DESIGN.md section 0 pastes this output, so that every number there can be
checked against the record by its key.
"""
import json
import sys


def load(path):
    txt = open(path).read()
    try:
        d = json.loads(txt)
        if isinstance(d, dict) and "parsed" in d:
            d = d["parsed"]
        if isinstance(d, dict) and "metric" in d:
            return d
    except ValueError:
        pass
    last = None
    for line in txt.splitlines():
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            try:
                last = json.loads(line)
            except ValueError:
                pass
    if last is None:
        raise SystemExit(f"{path}: no bench.py JSON line found")
    return last


def get(d, key):
    cur = d
    for part in key.split("."):
        if not isinstance(cur, dict) or part not in cur:
            return None
        cur = cur[part]
    return cur


def fmt(v, nd=3):
    if v is None:
        return "-"
    if isinstance(v, float):
        return f"{v:.{nd}f}".rstrip("0").rstrip(".") if abs(v) < 1000 else f"{v:.0f}"
    return str(v)


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    d = load(sys.argv[1])
    wl = get(d, "config.workload")
    rows = []

    def row(workload, what, ms_key=None, frac_key=None, of=None, value_key=None, unit=""):
        ms = get(d, ms_key) if ms_key else None
        frac = get(d, frac_key) if frac_key else None
        val = get(d, value_key) if value_key else None
        if ms is None and frac is None and val is None:
            return
        src = ", ".join(k for k in (value_key, ms_key, frac_key) if k)
        rows.append((workload, what, fmt(val) + (" " + unit if val is not None and unit else "") if val is not None else "-",
                     fmt(ms), fmt(frac), of or "-", src))

    row(wl, "headline: cost volumes / s through a1..a10", "ms_per_step", None, None, "value", "cv/s")
    row(wl, "plane_sweep_variance_kernel", "roofline.kernel_ms", "roofline.frac", "HBM 8 TB/s, algorithmic bytes")
    row(wl, "  same, against its store pattern alone", None, "roofline.frac_of_store_pattern_ceiling", "store-pattern probe on this box")
    row(wl, "  plane_sweep_coords_kernel (side stream)", "roofline.table_kernel_ms")
    for k in ("pack", "plane_sweep_geometry (side stream, beside the packing)", "plane_sweep_variance", "depth_prob_topk", "backproject_mean"):
        if get(d, "stage_ms") and k in d["stage_ms"]:
            rows.append((wl, f"  stage: {k}", "-", fmt(d["stage_ms"][k]), "-", "-", f"stage_ms[{k!r}]"))
    r = "reference_true_shape"
    row(get(d, r + ".workload"), "cost volumes / s", None, None, None, r + ".cost_volumes_per_sec", "cv/s")
    row(get(d, r + ".workload"), "plane_sweep_variance_kernel", r + ".sweep_kernel_ms", r + ".frac", "HBM 8 TB/s")
    row(get(d, r + ".workload"), "  same, against its store pattern alone", None, r + ".frac_of_store_pattern_ceiling", "store-pattern probe")
    for name, o in (get(d, "other_workloads") or {}).items():
        rows.append((name, "plane_sweep_variance_kernel", fmt(o.get("cost_volumes_per_sec")) + " cv/s", fmt(o.get("sweep_kernel_ms")),
                     fmt(o.get("frac")), "HBM 8 TB/s", f"other_workloads.{name}"))
        if o.get("frac_of_store_pattern_ceiling") is not None:
            rows.append((name, "  same, against its store pattern alone", "-", "-", fmt(o["frac_of_store_pattern_ceiling"]),
                         "store-pattern probe", f"other_workloads.{name}.frac_of_store_pattern_ceiling"))
    c = "with_cost_network"
    cw = get(d, c + ".workload")
    row(cw, "chain a1..a10 + cost network + neck + head, one stream", c + ".ms_per_scene", None, None, c + ".scenes_per_sec", "scenes/s")
    mmm = get(d, c + ".ms_per_scene_min_median_max")
    if mmm:
        rows.append((cw, "  same, per scene (events behind every scene): min / median / max", "-", " / ".join(fmt(v) for v in mmm), "-", "-",
                     c + ".ms_per_scene_min_median_max"))
        na = get(d, c + ".device_allocations_in_timed_loop")
        if na is not None:
            rows.append((cw, "  same, hipMalloc calls of the caching allocator inside the ten timed scenes", str(na), "-", "-", "-",
                         c + ".device_allocations_in_timed_loop"))
    row(cw, "  same, detector on the side stream", c + ".detector_on_side_stream.ms_per_scene", None, None, c + ".scenes_per_sec_pipelined", "scenes/s")
    row(cw, "  CostRegNet_3DGS forward (eval)", c + ".cost_network_roofline.network_ms", c + ".cost_network_roofline.network_vs_fp32_mfma_peak",
        "fp32 MFMA peak 157 TFLOP/s, useful FLOP")
    prec = get(d, c + ".cost_network_roofline.conv0_precision") or "bf16x3"
    row(cw, f"  conv0 256->64 ({'conv3d_k3_fp16mx_kernel' if prec == 'fp16mx' else 'conv3d_k3_bf16x3_kernel'})", c + ".cost_network_roofline.kernel_ms",
        c + ".cost_network_roofline.frac", "dense bf16 2.5 PFLOP/s, " + ("11/7 matrix-pipe units per product" if prec == "fp16mx" else "3 MFMAs per product"))
    by = get(d, c + ".cost_network_roofline.conv0_ms_by_route")
    if by:
        rows.append((cw, "  conv0 by route (the same tensor, same run)", "-", " / ".join(f"{k} {fmt(v)}" for k, v in by.items()), "-", "-",
                     c + ".cost_network_roofline.conv0_ms_by_route"))
    rt = get(d, c + ".detector_on_side_stream.route")
    if rt:
        rows.append((cw, "  route kept by overlap_detector = 'auto' (measured periods, ms)", rt, " / ".join(f"{k} {fmt(v)}" for k, v in (get(d, c + ".detector_on_side_stream.route_periods_ms") or {}).items()), "-", "-",
                     c + ".detector_on_side_stream.route"))
    row(cw, "  IndoorImVoxelNeck forward", c + ".neck_roofline.kernel_ms", c + ".neck_roofline.frac", "dense bf16 2.5 PFLOP/s, 3 MFMAs per product")
    row(cw, "  neck + head per scene at batch 4", c + ".detector_batch4.ms_per_scene")
    for name, o in (get(d, "with_cost_network_test_shapes") or {}).items():
        rows.append((name, "chain, one stream", fmt(o.get("scenes_per_sec")) + " scenes/s", fmt(o.get("ms_per_scene")), "-", "-",
                     f"with_cost_network_test_shapes.{name}"))
        rows.append((name, "  detector on the side stream" + (f" (route {o['pipelined_route']})" if o.get("pipelined_route") else ""),
                     fmt(o.get("scenes_per_sec_pipelined")) + " scenes/s", fmt(o.get("ms_per_scene_pipelined")),
                     "-", "-", f"with_cost_network_test_shapes.{name}"))
        rows.append((name, "  CostRegNet_3DGS forward", fmt(o.get("network_useful_TFLOPs")) + " useful TFLOP/s", fmt(o.get("network_ms")), "-", "-",
                     f"with_cost_network_test_shapes.{name}.network_ms"))
    t = "training"
    tw = get(d, t + ".workload")
    row(tw, "training step (AdamW + clip 35), stand-in cost network", t + ".stand_in_cost_network.ms_per_step", None, None, t + ".stand_in_cost_network.scenes_per_sec", "scenes/s")
    row(tw, "  of it: clip_grad_norm_ with its host read + AdamW", t + ".stand_in_cost_network.optimizer_ms")
    row(tw, "training step (AdamW + clip 35), real CostRegNet_3DGS (bf16x3 gradients: 1e-3 element-wise / 1e-4 in norm of the reference's)", t + ".real_cost_network.ms_per_step", None, None, t + ".real_cost_network.scenes_per_sec", "scenes/s")
    row(tw, "  of it: clip_grad_norm_ with its host read + AdamW", t + ".real_cost_network.optimizer_ms")
    row(tw, "  plane_sweep_variance_bwd (whole operator)", t + ".roofline.backward_sweep.kernel_ms", t + ".roofline.backward_sweep.frac", "HBM 8 TB/s")
    row(tw, "  conv0 weight gradient (bf16x3)", t + ".roofline.weight_gradient_conv0.kernel_ms", t + ".roofline.weight_gradient_conv0.frac", "dense bf16 2.5 PFLOP/s")
    row(tw, "  stride-2 weight gradient (bf16x3)", t + ".roofline.weight_gradient_stride2.kernel_ms", t + ".roofline.weight_gradient_stride2.frac", "dense bf16 2.5 PFLOP/s")
    row(wl, "CPU baseline (" + fmt(get(d, "cpu_baseline.kind")) + ", " + fmt(get(d, "cpu_baseline.cores")) + " cores)", None, None, None,
        "cpu_baseline.value", fmt(get(d, "cpu_baseline.unit")))
    print("| workload | kernel / quantity | value | ms | frac | of | key in the record |")
    print("|---|---|---|---|---|---|---|")
    for r_ in rows:
        print("| " + " | ".join(str(c_) for c_ in r_) + " |")


if __name__ == "__main__":
    main()
