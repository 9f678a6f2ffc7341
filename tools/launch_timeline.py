"""Start / end of every k_path_wavefront and k_reconstruct launch of a rocprofv3 kernel trace, relative to the first one:
how the three batch slots overlap inside a frame.     python tools/launch_timeline.py KERNEL_TRACE.csv [MAX_ROWS]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_path_wavefront" in r["Kernel_Name"] or "k_reconstruct" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 200]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = "path " if "k_path" in r["Kernel_Name"] else "recon"
    print(f"{name} q{r.get('Queue_Id', '?'):>3} start {a:9.3f} ms  end {b:9.3f} ms  dur {b - a:8.3f} ms")
