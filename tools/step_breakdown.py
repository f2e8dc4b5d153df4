"""Per-step time by kernel family from a rocprofv3 --kernel-trace --stats CSV of `bench.py --no-extra --no-roofline
--no-cpu-baseline` (one workload per CSV).    python tools/step_breakdown.py <kernel_stats.csv> <out.json> [label] [T]

T = timesteps of the recurrent encoders (the padded dialogue length): with it the GRU recurrence -- the family that
dominates the BASELINE cfg2-cfg4 steps and is bound by neither HBM nor the matrix pipe -- gets a stated bound: the serial
chain of T timesteps per launch, priced (a) by the measured floor of the ablation skeleton (the time loop with nothing but
the h write and the per-step barrier: 0.185 us per timestep, profiles/r03_gru_kernels.md) and (b) by the dependent chain of
a full timestep (barrier release -> h broadcast load -> 13 FMA groups -> 2-level add -> DPP add -> r, z, n gates -> h ->
LDS write: ~20 dependent VALU results at ~9 cycles, 4 transcendentals at ~25, 3 LDS latencies at ~100 and 330 cycles of
FMA issue = ~0.41 us at 2.2 GHz)."""
import csv
import json
import re
import sys

FAMILIES = [
    ("gru_recurrence", r"gru_seq_|gru_tab_"),
    ("weight_gradients", r"gemm_tn"),
    ("gcn_stack_fused", r"lstm_gate_|gcnii_layer_|gcn_input_|lstm_pointwise|gcnii_combine|prop_layer_strip"),
    ("propagate_K6", r"propagate_"),
    ("adjacency_K5_K6b", r"tile_dot|unit_cross|rdeg_cross|scale_tiles|symmetrize|bwd_rowsum|bwd_etile|bwd_ecross|cross_dot|unit_bwd|adj_strip|adj_finish"),
    ("projections_hand_written", r"linear_kernel|linear_split|linear_small|linear_lds|linear2|linear_planes|cut_planes"),
    ("fusion_modules", r"softmax_scale|mfn_mem|gated_pair|rowscale_colsum"),
    ("library_gemm_k_not_multiple_of_4", r"^Cijk_"),
    ("head_and_loss", r"head_|focal_loss"),
    ("encoder_glue", r"party_|mask_scale|colsum|keep_flags"),
    ("optimizer", r"adam_step"),
    ("aten_and_runtime", r"at::native|rocclr|rocprim|elementwise_kernel_with_index"),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    calls = {r["Name"]: int(r["Calls"]) for r in rows}
    # two forward recurrence launches per step (one per GRU layer), whichever kernel variants they ran on
    steps = sum(c for n, c in calls.items() if "gru_seq_fwd" in n) // 2
    if steps <= 0:
        raise SystemExit("no gru_seq_fwd kernel in the trace: cannot tell the step count")
    fam = {k: dict(us_per_step=0.0, launches_per_step=0.0) for k, _ in FAMILIES}
    fam["other"] = dict(us_per_step=0.0, launches_per_step=0.0)
    for r in rows:
        name = r["Name"]
        for k, pat in FAMILIES:
            if re.search(pat, name):
                break
        else:
            k = "other"
        fam[k]["us_per_step"] += float(r["TotalDurationNs"]) / 1e3 / steps
        fam[k]["launches_per_step"] += int(r["Calls"]) / steps
    total = sum(v["us_per_step"] for v in fam.values())
    out = {"source": sys.argv[1].split("/")[-1], "label": sys.argv[3] if len(sys.argv) > 3 else "", "steps_in_trace": steps,
           "kernel_us_per_step": round(total, 1), "launches_per_step": round(sum(v["launches_per_step"] for v in fam.values()), 1),
           "families": {k: {"us_per_step": round(v["us_per_step"], 1), "share": round(v["us_per_step"] / total, 3),
                            "launches_per_step": round(v["launches_per_step"], 1)}
                        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us_per_step"]) if v["us_per_step"] > 0}}
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    g = out["families"].get("gru_recurrence")
    if T and g:
        launches = g["launches_per_step"]
        floor_us, chain_us = 0.185, 0.41
        out["families"]["gru_recurrence"]["bound"] = {
            "kind": "latency: T serial timesteps per launch (no HBM / MFMA roofline applies)", "T": T,
            "recurrence_launches_per_step": launches,
            "skeleton_floor_us_per_timestep": floor_us, "bound_us_per_step": round(launches * T * floor_us, 1),
            "frac": round(launches * T * floor_us / g["us_per_step"], 3),
            "dependent_chain_us_per_timestep": chain_us, "chain_bound_us_per_step": round(launches * T * chain_us, 1),
            "chain_frac": round(launches * T * chain_us / g["us_per_step"], 3),
            "achieved_us_per_timestep": round(g["us_per_step"] / (launches * T), 3),
            "source": "profiles/r03_gru_kernels.md (compile-time ablations of gru_seq_fwd_io_kernel)"}
    dom = next(iter(out["families"]))
    out["dominant"] = "%s: %.0f us of %.0f us kernel time per step (%.0f %%)" % (
        dom, out["families"][dom]["us_per_step"], total, 100 * out["families"][dom]["share"])
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
