"""CPU placement of the launching process (no torch, no HIP: import this FIRST).

The dependent chain of launches is sensitive to where the launching thread and the HIP runtime's helper threads run: about one fresh process in five
went 10-11 % slower as a whole when they shared or changed cores ("slow mode"; tools/slowmode.sh: 0 of 20 runs under
`taskset -c <physical cores of one socket>`).  An affinity set after the runtime's threads exist does not confine them, so the mask has to be in place
BEFORE HIP initialises: pin_for_gpu() confines the calling thread — and every thread created after it — to the physical cores (one hardware thread per
core) of the NUMA node the GPU hangs off.  PBWTAMD_PIN=0 opts out.  The C library does the same in pbwtamd_engine_create when it is the first HIP user."""
import glob
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(text):
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device=0):
    """NUMA node of the device-th AMD GPU (PCI vendor 0x1002 under /sys/class/drm), or -1"""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(dev, "vendor")) == "0x1002":
            cards.append(dev)
    if not cards:
        return -1
    node = _read(os.path.join(cards[min(device, len(cards) - 1)], "numa_node"))
    try:
        return int(node)
    except (TypeError, ValueError):
        return -1


def physical_cores(node=-1):
    """one hardware thread per core among the CPUs of `node` (all online CPUs for node < 0)"""
    cpus = _cpulist(_read("/sys/devices/system/node/node%d/cpulist" % node)) if node >= 0 else []
    if not cpus:
        cpus = _cpulist(_read("/sys/devices/system/cpu/online")) or list(range(os.cpu_count() or 1))
    keep = []
    for c in cpus:
        sib = _cpulist(_read("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c))
        if not sib or min(sib) == c:
            keep.append(c)
    return keep or cpus


def pin_for_gpu(device=0):
    """returns the CPU set now in force (None: opted out or not possible)"""
    if os.environ.get("PBWTAMD_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        want = set(physical_cores(gpu_numa_node(device))) & allowed
        if len(want) < 2:                                   # (a container with a narrow cpuset: leave it alone)
            return None
        os.sched_setaffinity(0, want)
        return sorted(want)
    except OSError:
        return None
