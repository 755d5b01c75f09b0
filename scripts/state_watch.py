"""Run a command while sampling every numeric field of the SMU's metrics
table (amdsmi_get_gpu_metrics_info) with wall-clock stamps, and write both
side by side: which field moves when the store pattern changes speed
(scripts/labsrc/state_lab.hip)?

    python scripts/state_watch.py out.json -- scripts/labsrc/state_lab 30 12 2
"""
import json
import subprocess
import sys
import threading
import time


def flatten(m, prefix=""):
    out = {}
    for k, v in m.items():
        if isinstance(v, bool):
            continue
        if isinstance(v, (int, float)):
            out[prefix + k] = float(v)
        elif isinstance(v, (list, tuple)):
            nums = [float(x) for x in v if isinstance(x, (int, float))
                    and not isinstance(x, bool)]
            nums = [x for x in nums if x < 4e9]     # 0xffff...: not populated
            if nums:
                out[prefix + k + ".mean"] = sum(nums)/len(nums)
                out[prefix + k + ".max"] = max(nums)
        elif isinstance(v, dict):
            out.update(flatten(v, prefix + k + "."))
    return out


def main():
    dst, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    samples, stop, err = [], threading.Event(), [None]

    def sampler():
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[0]
            while not stop.is_set():
                t = time.time()
                row = flatten(amdsmi.amdsmi_get_gpu_metrics_info(h))
                row["t"] = t
                samples.append(row)
                time.sleep(0.02)
        except Exception as e:      # noqa: BLE001 -- reported, not fatal
            err[0] = repr(e)[:300]
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(0.5)
    res = subprocess.run(cmd, capture_output=True, text=True)
    time.sleep(0.3)
    stop.set()
    th.join(timeout=2)
    lines = []
    for line in res.stdout.splitlines():
        try:
            lines.append(json.loads(line))
        except ValueError:
            pass
    # keep the fields that ever change, plus a few that matter either way
    keys = sorted({k for s in samples for k in s})
    moving = [k for k in keys if len({s.get(k) for s in samples}) > 1]
    json.dump({"cmd": cmd, "rc": res.returncode, "stderr": res.stderr[-500:],
               "sampler_error": err[0], "lab": lines, "fields": moving,
               "constant": {k: samples[0].get(k) for k in keys
                            if k not in moving} if samples else {},
               "samples": [[s.get(k) for k in moving] for s in samples]},
              open(dst, "w"))
    print("lab lines %d, samples %d, moving fields %d, sampler error %s"
          % (len(lines), len(samples), len(moving), err[0]))


if __name__ == "__main__":
    main()
