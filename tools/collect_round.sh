#!/bin/bash
# Turns the output directories of tools/profile_round.sh (S-longdress, S-owlii) and tools/profile_smooth.sh into the
# files of profiles/rNN/.  Usage: tools/collect_round.sh <round> <prof longdress> <prof owlii> <prof smooth> [driver bench json]
set -e
r=$1; ld=$2; ow=$3; sm=$4; drv=$5
out=profiles/r$(printf %02d $r); mkdir -p $out
python3 tools/traffic_json.py $ld $r k_recon_tiles "S-longdress, 128 frames per launch (4 GOFs of 32)" > $out/traffic.json
python3 tools/traffic_json.py $ow $r k_recon_tiles "S-owlii, 128 frames per launch (8 distinct frames x 16)" > $out/traffic_owlii.json
python3 tools/traffic_json.py $sm $r k_smooth "S-longdress, 128 frames per launch (4 GOFs of 32), geometry + colour smoothing: every k_smooth_* launch of a step" > $out/traffic_smooth.json
cp "$(ls -t $ld/stats/*/*_kernel_stats.csv | head -1)" $out/longdress_kernel_stats.csv
cp "$(ls -t $ow/stats/*/*_kernel_stats.csv | head -1)" $out/owlii_kernel_stats.csv
cp "$(ls -t $sm/stats/*/*_kernel_stats.csv | head -1)" $out/smooth_kernel_stats.csv
grep '^{"metric"' $ld/stats.log | tail -1 > $out/bench_longdress.json
grep '^{"metric"' $ow/stats.log | tail -1 > $out/bench_owlii.json
grep '^{"metric"' $sm/stats.log | tail -1 > $out/bench_smooth.json
[ -n "$drv" ] && grep '^{"metric"' $drv | tail -1 > $out/bench_driver_command.json
python3 - $out <<'PY'
import json, sys, csv
out = sys.argv[1]
for f in ("traffic", "traffic_owlii", "traffic_smooth"):
    d = json.load(open(f"{out}/{f}.json"))
    print(f"{f:16s} read {d['hbm_read_bytes_per_launch']/1e6:8.1f} MB  written {d['hbm_write_bytes_per_launch']/1e6:8.1f} MB  total {d['hbm_bytes_per_launch']/1e6:8.1f} MB  sources {d['kernel_source_sha16']}")
for f in ("longdress", "owlii", "smooth", "driver_command"):
    try:
        d = json.load(open(f"{out}/bench_{f}.json"))
    except Exception as e:
        print(f, "missing", e); continue
    r = d["roofline"]
    print(f"{f:16s} ms_per_step {d['ms_per_step']}  kernel_ms {r['kernel_ms']}  frac {r['frac']}  frac_traffic {r.get('frac_traffic')}  stale {r.get('traffic_stale')}")
for f in ("longdress", "owlii", "smooth"):
    for row in csv.DictReader(open(f"{out}/{f}_kernel_stats.csv")):
        if "vpcc" in row["Name"]:
            print(f"{f:10s} {row['Name'].split('(')[0][-40:]:40s} calls {row['Calls']:>5s}  avg {float(row['AverageNs'])/1e3:9.1f} us")
PY
