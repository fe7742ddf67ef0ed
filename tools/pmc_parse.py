import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("eval_rows_mdelta_kernel", "eval_rows_delta_kernel", "eval_rows_f32_kernel", "eval_rows_bf16_kernel", "td_trace_kernel", "td_forward_kernel", "stage2_kernel<3>", "leaves_kernel", "roots_kernel", "expand_kernel<3>", "expand_all_kernel", "doubles_kernel", "boundary_kernel<true>", "boundary_kernel<false>", "root_hidden_bf16x3_kernel", "root_hidden_resident_kernel", "apply_kernel", "step_random_kernel"):
        if key in n:
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[-10:]
        print("   %-32s %.4g" % (c, sum(v) / len(v)))
