"""Sample socket power (hwmon power1_input) and shader clock (freq1_input) of every GPU in sysfs every 10 ms for <seconds>; print the mean
over the busiest card's busy samples.  Started next to a kernel loop by tools/power_probe.sh."""
import glob, sys, time
secs, tag = float(sys.argv[1]), sys.argv[2]
cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
def rd(p):
    try:
        return float(open(p).read())
    except Exception:
        return float("nan")
samples = {c: [] for c in cards}
t0 = time.time()
while time.time() - t0 < secs:
    for c in cards:
        samples[c].append((rd(c + "/power1_input") / 1e6, rd(c + "/freq1_input") / 1e6))
    time.sleep(0.01)
best = max(cards, key=lambda c: sum(p for p, _ in samples[c]) / max(1, len(samples[c])))
s = samples[best]
s = s[len(s) // 4: len(s) * 9 // 10]                    # skip the ramp at the start and the tail
pw = [p for p, _ in s]; fr = [f for _, f in s]
cap = rd(best + "/power1_cap") / 1e6
print(f"{tag:58s} power mean {sum(pw) / len(pw):7.1f} W  max {max(pw):7.1f} W (cap {cap:.0f} W)   sclk mean {sum(fr) / len(fr):6.0f} MHz  min {min(fr):6.0f}  max {max(fr):6.0f}   ({len(s)} samples, {best.split('/')[4]})", flush=True)
