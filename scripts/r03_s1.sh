#!/bin/bash
# round 3, GPU session 1: what the box offers for clocks/power, the GPU suite
# as it stands, the row-stride sweep, a default bench line
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s1
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== rocm-smi"; timeout 60 rocm-smi --showclocks --showpower --showtemp --showperflevel --json 2>&1 | head -c 6000
  echo; echo "== amd-smi metric"; timeout 60 amd-smi metric --json 2>&1 | head -c 12000
  echo; echo "== sysfs"; ls /sys/class/drm/ 2>&1; for d in /sys/class/drm/card*/device; do echo $d; ls $d | head -80; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk; do echo "-- $f"; cat $d/$f 2>&1; done; ls $d/hwmon/*/ 2>&1 | head -60; done
  echo "== python amdsmi"; timeout 60 python - <<'PY'
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print(len(hs), "handles")
    h = hs[0]
    for name in ("amdsmi_get_gpu_metrics_info", "amdsmi_get_power_info", "amdsmi_get_clock_info"):
        try:
            fn = getattr(amdsmi, name)
            if name == "amdsmi_get_clock_info":
                for ct in ("GFX", "MEM", "SYS", "DF", "DCEF", "SOC"):
                    try:
                        print(name, ct, fn(h, getattr(amdsmi.AmdSmiClkType, ct)))
                    except Exception as e:
                        print(name, ct, "ERR", repr(e)[:200])
            else:
                print(name, fn(h))
        except Exception as e:
            print(name, "ERR", repr(e)[:300])
    amdsmi.amdsmi_shut_down()
except Exception as e:
    print("amdsmi failed", repr(e)[:300])
PY
} > $OUT/smi_probe.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 600 python scripts/r03_ldpad.py > $OUT/ldpad.jsonl 2> $OUT/ldpad.err
tail -3 $OUT/ldpad.jsonl
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
head -c 1500 $OUT/bench.json
