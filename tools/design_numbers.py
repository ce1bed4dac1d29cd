#!/usr/bin/env python3
"""Prints the "Measured" table of DESIGN.md section 4 from an evidence set under profiles/ (bench line + counter files).
usage: tools/design_numbers.py r06_n"""
import csv
import json
import sys

tag = sys.argv[1]
P = "profiles/" + tag + "_"
d = json.load(open(P + "bench.json"))
tr = {w: json.load(open(P + w + "_traffic.json")) for w in ("dense", "slots", "modular")}
gb = lambda t, pre="": sum(v["hbm_bytes"] for k, v in t.items() if isinstance(v, dict) and k.startswith(pre)) / 1e9
r, e, s = d["roofline"], d["e2e_pcie_inclusive"], d["secondary"]
k1, fl, fa, one = d["k1"], d["filters"], d["filters_all_blocks_filtered"], d["one_frame_in_flight"]
half = r["epf_population_half_active"]["k23_fused_filters"]
sl = e["slots_resident_no_pcie"]
ks = {}
for row in csv.DictReader(open(P + "slots_kernel_stats.csv")):
    for key in ("k1_scan", "k1_dct8<3", "k1_dct16_32<3", "k1_dct16_32<2"):
        if key in row["Name"]:
            ks[key] = float(row["AverageNs"]) / 1e3
kd = {}
for row in csv.DictReader(open(P + "dense_kernel_stats.csv")):
    for key in ("k1_scan", "k1_dct8<0", "k1_dct16_32<0"):
        if key in row["Name"]:
            kd[key] = float(row["AverageNs"]) / 1e3
npx = 8192 * 8192
c2, c4, c5 = s["config2_4096_d1_epf0"], s["config4_modular_8192"], s["config5_16384_all_types"]
c53 = c5["epf_iters_3"]
p16, p12 = e["slots_pos6_val10_no_sort"], e["slots_packed12_no_sort"]
ho = lambda leg: sorted(leg["both_upload_orderings"]["host_ordered_1_streams_ms"])[1]
sw = e["slot_form_content_sweep"]
print(f"""| what | ms | against |
|---|---|---|
| **`value`: 8192² d1, dense i32 slabs, two frames in flight** | **{d['ms_per_step']:.4f} per frame = {d['value'] / 1e3:.1f} GP/s** (one frame in flight {one['wall_ms_per_frame']:.3f}; sum of kernels {one['sum_of_kernels_ms']:.3f}: gap {one['gap_ms']:.3f}) | fused-ideal 24.4 B/px: {r['chain_vs_fused_ideal']['frac_pipelined']:.2f} of 8 TB/s; counter bytes {gb(tr['dense']):.2f} GB: {d['chain_counter_frac_of_8TBs']:.2f} of 8 TB/s, {d['chain_counter_frac_of_6p3TBs']:.2f} of 6.3 |
| K1, dense (scan {kd.get('k1_scan', 0):.1f} us + dct8 {kd.get('k1_dct8<0', 0):.0f} + dct16_32 {kd.get('k1_dct16_32<0', 0):.0f}) | {k1['ms']:.3f} | 24.3 B/px: **{k1['frac']:.3f}** of 8 TB/s; counters {gb(tr['dense'], 'k1_'):.3f} GB = {gb(tr['dense'], 'k1_') * 1e9 / k1['algorithmic_bytes']:.2f}x algorithmic: {gb(tr['dense'], 'k1_') / k1['ms']:.1f} TB/s |
| filters (Gaborish + EPF1 + EPF2), spec population | {fl['ms']:.3f} | 24.06 B/px: **{fl['frac']:.3f}**; counters {gb(tr['dense'], 'k23'):.3f} GB = {gb(tr['dense'], 'k23') * 1e9 / fl['algorithmic_bytes']:.2f}x: {gb(tr['dense'], 'k23') / fl['ms']:.1f} TB/s |
| filters, half / all blocks filtered | {half['ms_per_step']:.3f} / {fa['ms']:.3f} | {half['frac']:.3f} / {fa['frac']:.3f} |
| slot-bucketed entries resident (the transport's form) | **{sl['ms_per_frame']:.3f} per frame = {sl['value'] / 1e3:.1f} GP/s**; K1 {sl['kernels_ms']['k1_vardct']:.4f} (scan {ks.get('k1_scan', 0):.0f} + dct8 {ks.get('k1_dct8<3', 0):.0f} + dct16_32 {ks.get('k1_dct16_32<3', 0):.0f} + fallback launch {ks.get('k1_dct16_32<2', 0):.0f} us) | chain counters {gb(tr['slots']):.2f} GB; K1 {gb(tr['slots'], 'k1_'):.2f} GB |
| PCIe-inclusive, 16-bit / 12-bit entries (device-ordered uploads, one slot stream per context) | {p16['ms_per_frame']:.3f} / {p12['ms_per_frame']:.3f} (host-ordered: {ho(p16):.3f} / {ho(p12):.3f}) | + host pack {sw['host_pack']['host_pack_ms_per_frame']:.1f} ms per frame on {sw['host_pack']['cores']} cores if the decoder keeps dense slabs |
| config 2: 4096² d1, EPF off | {c2['ms_per_step']:.3f} (K1 {c2['kernels']['k1_vardct']['ms_per_step']:.3f} = {c2['kernels']['k1_vardct']['frac']:.2f}, Gaborish {c2['kernels']['k23_fused_filters']['ms_per_step']:.3f} = {c2['kernels']['k23_fused_filters']['frac']:.2f}) | |
| config 4: 8192² x 3 Modular chain + RCT | {c4['chain']['ms']:.3f} (level by level {c4['chain']['level_by_level_ms']:.3f}); palette {c4['palette']['ms']:.3f} = {c4['palette']['frac']:.2f}; RCT alone {c4['rct_alone']['ms']:.3f} = {c4['rct_alone']['frac']:.2f} | 16 B per final sample: {c4['chain']['frac']:.3f}; counters {gb(tr['modular']):.2f} GB |
| config 5: 16384² all 27 types | {c5['ms_per_step']:.2f} (K1 {c5['kernels']['k1_vardct']['ms_per_step']:.2f} = {c5['kernels']['k1_vardct']['frac']:.3f}, filters {c5['kernels']['k23_fused_filters']['ms_per_step']:.2f} = {c5['kernels']['k23_fused_filters']['frac']:.3f}); `epf_iters = 3`: {c53['ms_per_step']:.2f} (filters {c53['kernels']['k23_fused_filters']['ms_per_step']:.2f} = {c53['kernels']['k23_fused_filters']['frac']:.3f}) | |
| CPU oracle (scalar C, {d['cpu_baseline']['cores']} cores) | {d['cpu_baseline']['value']:.0f} MP/s | reported baseline, not the reference's SIMD path |
""")
f = lambda k: f"{sw[k]['slots_resident_ms_per_frame']:.3f} / {sw[k]['k1_ms']:.3f}"
print(f"""Content sweep of the slot form in the same run (`slot_form_content_sweep`; slot-resident frame / K1 ms; dense-resident frame
{sw['d1_clean']['dense_resident_ms_per_frame']:.3f} and dense K1 {sw['d1_clean']['dense_k1_ms']:.3f} at every density): clean {f('d1_clean')} -- outliers (1e-5 of the entries at ±2000…30000,
split by the packer, {100 * sw['outliers_1e-5_of_entries_2000_to_30000']['fallback_share_of_batches']:.1f} % of the batches fall back) {f('outliers_1e-5_of_entries_2000_to_30000')} ({100 * (sw['outliers_1e-5_of_entries_2000_to_30000']['vs_clean_slots_frame'] - 1):+.0f} %) -- density x0.5 {f('density_x0.5')} -- x2 {f('density_x2')}
({100 * sw['density_x2']['fallback_share_of_batches']:.0f} % of the batches fall back; {sw['density_x2']['slots_vs_dense']:.2f}x the dense-resident frame) -- x4 {f('density_x4')} (every batch; {sw['density_x4']['slots_vs_dense']:.2f}x) -- one group
as a dense slab {f('one_group_dense_slab')} ({100 * (sw['one_group_dense_slab']['vs_clean_slots_frame'] - 1):+.0f} %).  Copy probe of the box: {d['copy_ceiling_GBs'] / 1e3:.2f} TB/s.""")
