"""A floor for every kernel family of one training step: what THIS decomposition of the MM-DFN step can
reach on an MI355X, family by family, so the distance of the measured step from it is a number.

For every launch class of a family (the launches the step issues, with the shapes it issues them at):

    floor(class) = count x max( algorithmic HBM bytes / 8 TB/s ,
                                useful flops / matrix peak ,
                                serial chain (only the GRU recurrence: T dependent timesteps x the dependent chain of one) ,
                                t_launch )

`t_launch` is the cost of one more DEPENDENT launch inside the replayed hipGraph -- measured live by bench.py (a captured chain
of dependent one-workgroup kernels; MI355X_MICROARCH.md price list "boundary": 1.45 us between trivial kernels, 1.7-1.9 between
streaming ones) -- and launches of a family that carry no dense work (normalisation chains, fills, index glue) are priced at
t_launch or their bytes, whichever is larger.  Matrix peak: the package computes fp32 products either on exact-f32 MFMAs (157.3
TFLOP/s) or as six bf16 piece products (2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32 flops); the floor uses the FASTER of the two, so
it is a lower bound for either arithmetic (the exact-f32 figure is reported next to it as `floor_f32_mfma_us`).

The inventory below is analytic (shapes from the workload: B dialogues, padded length L, true lengths, widths, layers, speakers)
and mirrors what the step launches; `launches` of the measured trace (tools/step_breakdown.py) is reported next to the count of
classes here so a drift between the two is visible.  Bytes are ALGORITHMIC (each operand once), like SURVEY 8d's K6 figure.
"""
import json
import sys

HBM = 8.0e12
F32_MFMA = 157.3e12
BF16_PIECE = 2.5e15 / 6.0
GRU_CHAIN_US = 0.41          # dependent chain of one timestep of gru_seq_fwd_io_kernel (profiles/r03_gru_kernels.md, model)
GRU_MFMA_STEP_US = 0.96      # matrix-pipe time of one MFMA-form step, 16 sequences per workgroup (profiles/r05_gru_mfma_form.md)


def _cls(name, count, byts=0.0, flops=0.0, serial_us=0.0):
    return dict(name=name, count=count, bytes=float(byts), flops=float(flops), serial_us=float(serial_us))


def inventory(B, L, lengths, D_t, D_a, D_v, nl, P, n_act=2, He=200, d=100, C=6, mfma_gru=False):
    """Launch classes per family for the MM-DFN configuration (speaker_weights '3-0-1': two modalities take the party GRU)."""
    M = 3
    R = L * B                               # padded utterance rows (the reference projects the padding too, model.py:1065)
    RP = n_act * P * R                      # party rows: (modality, speaker, dialogue) sequences of full length L
    N = sum(lengths)
    MN = M * N
    nnz = sum(M * l * l + M * (M - 1) * l for l in lengths)
    F = He
    G = 600                                 # gate rows of both directions
    Dsum = D_t + D_a + D_v
    seqs = B + n_act * P * B                # context + party sequences (each both directions) of one recurrence launch
    fam = {}

    def dense(rows, k, n):
        return dict(byts=4.0 * (rows * k + k * n + rows * n), flops=2.0 * rows * k * n)

    fam["projections_hand_written"] = [
        _cls("modality projections (3 groups)", 1, 4.0 * (R * Dsum + Dsum * He + 3 * R * He), 2.0 * R * Dsum * He),
        # (project-then-gather: the party branch projects its n_act x R utterance rows, not the P-fold party rows; the context
        # GRU's contraction rides in the same grouped launch)
        _cls("GRU l0 input contractions (party modalities + context, one grouped launch)", 1, **dense(n_act * R + R, He, G)),
        _cls("GRU l1 input contraction (context + party, one grouped launch)", 1, **dense(R + RP, He, G)),
        _cls("dX of GRU l1 (context + party, one grouped launch)", 1, **dense(R + RP, G, He)),
        _cls("dX of party GRU l0", 1, **dense(n_act * R, G, He)),
        _cls("dX of context GRU l0", 1, **dense(R, G, He)),
    ]
    T = L
    step_us = GRU_MFMA_STEP_US if mfma_gru else GRU_CHAIN_US
    gru_f = 4.0 * seqs * T * (G + He + 2 * 4 * 100)            # gi in, y out, four saved gate tensors (both directions)
    gru_b = 4.0 * seqs * T * (He + 2 * 4 * 100 + 2 * 100 + G)   # dy, saved gates, h_prev in; dgi out
    rec_flops = 2.0 * seqs * T * 2 * 300 * 100
    fam["gru_recurrence"] = [
        _cls("recurrence forward (l0, l1)", 2, gru_f, rec_flops, serial_us=T * step_us),
        _cls("recurrence backward (l1, l0)", 2, gru_b, 2 * rec_flops, serial_us=T * step_us),
    ]
    fam["gcn_stack_fused"] = [
        _cls("input layer forward (x.W0 + ReLU + dropout)", 1, 4.0 * MN * (2 * F + 2 * d), 2.0 * MN * F * d),
        _cls("input layer backward (dX)", 1, 4.0 * MN * (2 * F + 2 * d), 2.0 * MN * F * d),
        _cls("LSTM gate forward (K8)", nl, 4.0 * MN * 9 * d, 2.0 * MN * 2 * d * 4 * d),
        _cls("LSTM gate backward (K8)", nl, 4.0 * MN * 14 * d, 2.0 * MN * 4 * d * 2 * d),
        # short dialogues: propagate + layer update of a strip in ONE launch (csrc/gcn_small.hip)
        _cls("propagate + GCNII layer forward (K6 + K7, one launch)", nl, 4.0 * MN * 6 * d + 4.0 * nnz, 2.0 * MN * 2 * d * d + 2.0 * d * nnz),
        _cls("GCNII layer backward (K7)", nl, 4.0 * MN * 6 * d, 2.0 * MN * 2 * d * d),
    ]
    k6_b = 4.0 * nnz + 8.0 * MN * d
    fam["propagate_K6"] = [
        _cls("propagate backward (dH = A.dO)", nl, k6_b, 2.0 * d * nnz),
    ]
    tiles = sum(M * l * l for l in lengths)
    # short dialogues (L <= 128): the strip form of csrc/adjacency_small.hip -- forward two launches, backward one
    fam["adjacency_K5_K6b"] = [
        _cls("strip forward (unit rows, cross cosines, Gram, similarity, degrees)", 1, 8.0 * MN * F + 8.0 * nnz, 2.0 * F * tiles),
        _cls("finish (column degrees onto the tiles, cross diagonals)", 1, 8.0 * nnz),
        _cls("adjacency gradient of the stack (dA = dO.H^T, all layers)", 1, 8.0 * MN * nl * d + 4.0 * nnz, 2.0 * nl * d * nnz),
        _cls("strip backward (d(degree), E, E.unit, unit-vector backward)", 1, 12.0 * nnz + 12.0 * MN * F, 2.0 * F * nnz),
    ]
    wg = []
    for c in fam["projections_hand_written"][:3]:           # (the forward contractions: each has a weight gradient)
        wg.append((c["flops"], c["bytes"]))
    for c in fam["gcn_stack_fused"]:
        if "forward" in c["name"]:
            wg.append((c["flops"] * c["count"], 4.0 * MN * 6 * d * c["count"]))
    whh = 2.0 * 2 * 2 * (seqs * T) * 300 * 100              # dW_hh of 2 layers x 2 directions
    wg.append((whh, 4.0 * 4 * seqs * T * 400))
    fam["weight_gradients"] = [
        _cls("dW / db of every dense layer, GRU weight, LSTM gate and GCN layer (one launch)", 1,
             sum(b for _, b in wg), sum(f for f, _ in wg)),
        _cls("fixed-order reduction of the split partial sums", 1, 3 * 4.0e6),
    ]
    fam["encoder_glue"] = [
        _cls("party gather / scatter-combine, forward + backward (4 launches)", 4, 4.0 * (RP * He + R * He) * 1.0),
        _cls("dropout flags, the forward mask of the GRU inter-layer dropout (2 launches)", 2, 4.0 * R * He),
    ]
    fam["head_and_loss"] = [
        _cls("head forward (dropout, ReLU, smax_fc, log-softmax)", 1, 4.0 * N * (900 + C)),
        _cls("FocalLoss forward (+ its gradient)", 1, 4.0 * N * 2 * C),
        _cls("head backward", 1, 4.0 * N * (2 * 900 + C), 2.0 * 2 * N * 900 * C),
    ]
    fam["aten_and_runtime"] = [_cls("fills / copies left to the runtime", 2, 0.0)]
    fam["optimizer"] = []
    return fam


def floors(fam_inventory, t_launch_us, measured=None):
    """-> {family: {floor_us, floor_f32_mfma_us, classes: [...], launches_modelled, (us_per_step, frac)}}, step_floor_us"""
    out = {}
    total = total_f32 = 0.0
    for fam, classes in fam_inventory.items():
        rows = []
        f_sum = f32_sum = 0.0
        n = 0
        for c in classes:
            hbm_us = c["bytes"] / HBM * 1e6
            mfma_us = c["flops"] / BF16_PIECE * 1e6
            f32_us = c["flops"] / F32_MFMA * 1e6
            one = max(hbm_us, mfma_us, c["serial_us"], t_launch_us)
            one32 = max(hbm_us, f32_us, c["serial_us"], t_launch_us)
            binding = max((("hbm", hbm_us), ("matrix", mfma_us), ("serial chain", c["serial_us"]), ("launch", t_launch_us)),
                          key=lambda kv: kv[1])[0]
            rows.append(dict(name=c["name"], count=c["count"], hbm_us=round(hbm_us, 2), matrix_bf16_piece_us=round(mfma_us, 2),
                             matrix_f32_us=round(f32_us, 2), serial_us=round(c["serial_us"], 2), binding=binding,
                             floor_us=round(one * c["count"], 2)))
            f_sum += one * c["count"]
            f32_sum += one32 * c["count"]
            n += c["count"]
        ent = dict(floor_us=round(f_sum, 1), floor_f32_mfma_us=round(f32_sum, 1), launches_modelled=n, classes=rows)
        if measured and fam in measured:
            us = measured[fam]["us_per_step"]
            extra = max(0.0, measured[fam].get("launches_per_step", n) - n)
            ent["floor_us"] = round(f_sum + extra * t_launch_us, 1)          # launches of the trace the inventory has no class for
            ent["floor_f32_mfma_us"] = round(f32_sum + extra * t_launch_us, 1)
            ent["us_per_step"] = us
            ent["launches_per_step"] = measured[fam].get("launches_per_step")
            ent["frac"] = round(ent["floor_us"] / us, 3) if us else None
        total += ent["floor_us"]
        total_f32 += ent["floor_f32_mfma_us"]
        out[fam] = ent
    return out, total, total_f32


CONFIGS = {
    "cfg2": dict(B=16, L=110, D_t=100, D_a=100, D_v=512, nl=2, P=2, C=6),
    "cfg4": dict(B=32, L=110, D_t=100, D_a=100, D_v=512, nl=2, P=2, C=6),
    "cfg3": dict(B=32, L=33, D_t=600, D_a=300, D_v=342, nl=4, P=9, C=7),
}


def for_config(name, lengths=None, t_launch_us=1.45, measured=None):
    c = dict(CONFIGS[name])
    lengths = list(lengths) if lengths is not None else [c["L"]] * c["B"]
    seqdirs = 2 * (c["B"] + 2 * c["P"] * c["B"])
    inv = inventory(lengths=lengths, mfma_gru=seqdirs > 1024, **c)
    fam, total, total32 = floors(inv, t_launch_us, measured)
    return dict(method="tools/step_floor.py: per launch class max(algorithmic bytes / 8 TB/s, flops / (2.5 PF / 6 piece products), "
                       "serial chain, dependent-launch cost)", t_launch_us=t_launch_us, families=fam,
                step_floor_us=round(total, 1), step_floor_f32_mfma_us=round(total32, 1))


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    measured = None
    if len(sys.argv) > 2:
        measured = json.load(open(sys.argv[2]))["families"]
    r = for_config(name, measured=measured)
    for k, v in r["families"].items():
        print("%-28s floor %7.1f us (f32 MFMA %7.1f)  measured %s  frac %s" % (
            k, v["floor_us"], v["floor_f32_mfma_us"], v.get("us_per_step"), v.get("frac")))
    print("step floor %.1f us (exact-f32 MFMA arithmetic: %.1f us)" % (r["step_floor_us"], r["step_floor_f32_mfma_us"]))
