"""Host topology for one-process-per-GPU runs (the scaling risk of the sharded path is host-side).

Which cores should rank r use?  The ones of the NUMA node its GPU hangs off: the rank's copy threads then write the pinned arenas
(allocated after the binding, so first-touch places them on that node) and the GPU's DMA engine reads them without crossing the
socket interconnect.  Everything here reads sysfs only -- no HIP call, so it runs before the process opens the GPU -- and takes the
sysfs root as a parameter, so the mapping is unit-tested on a CPU box against a fake tree (tests/test_topology.py).

  GPU ordinal -> PCI address : KFD topology (/sys/class/kfd/kfd/topology/nodes/N/properties: simd_count > 0 = a GPU, `domain`,
                               `location_id` = bus << 8 | device << 3 | function), in node order = HIP's device order;
                               ROCR_VISIBLE_DEVICES filters first (the ROCr layer), then HIP_VISIBLE_DEVICES if it is set, otherwise
                               CUDA_VISIBLE_DEVICES (the HIP runtime reads one of the two, never both).
                               Fallback: /sys/class/drm/card*/device of vendor 0x1002, by card number.
  PCI address -> NUMA node   : /sys/bus/pci/devices/<address>/numa_node (-1: unknown)
  NUMA node   -> cores       : /sys/devices/system/node/node<N>/cpulist
"""
from __future__ import annotations

import glob
import os
import re


def parse_cpulist(text: str) -> list[int]:
    out: list[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def _read(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_pci_addresses(sysfs: str = "/sys") -> list[str]:
    """PCI addresses of the GPUs in HIP's device order (no visibility filter applied)"""
    gpus = []
    nodes = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*/properties"))
    for path in sorted(nodes, key=lambda p: int(os.path.basename(os.path.dirname(p)))):
        text = _read(path)
        if not text:
            continue
        props = dict(line.split(None, 1) for line in text.splitlines() if len(line.split(None, 1)) == 2)
        if int(props.get("simd_count", "0")) <= 0:
            continue                                    # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        gpus.append(f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}")
    if gpus:
        return gpus
    cards = []
    for path in glob.glob(os.path.join(sysfs, "class/drm/card*")):
        m = re.fullmatch(r"card(\d+)", os.path.basename(path))
        if not m or _read(os.path.join(path, "device/vendor")) != "0x1002":
            continue
        dev = os.path.realpath(os.path.join(path, "device"))
        cards.append((int(m.group(1)), os.path.basename(dev)))
    return [addr for _n, addr in sorted(cards)]


def visible_indices(n_gpus: int, env=os.environ) -> list[int]:
    """device ordinals the process sees, as indices into the unfiltered list (integer index lists only; UUID forms are ignored)"""
    idx = list(range(n_gpus))

    def apply(var: str) -> bool:
        """filter by one variable; False when it is unset / empty (the next candidate is looked at)"""
        nonlocal idx
        v = env.get(var)
        if v is None or v.strip() == "":
            return False
        try:
            sel = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return True                                 # a UUID list: set, but not ours to interpret
        idx = [idx[i] for i in sel if 0 <= i < len(idx)]
        return True

    apply("ROCR_VISIBLE_DEVICES")
    # the HIP runtime uses HIP_VISIBLE_DEVICES when it is set and CUDA_VISIBLE_DEVICES only otherwise: launchers that export both
    # with the same list must not be filtered twice
    if not apply("HIP_VISIBLE_DEVICES"):
        apply("CUDA_VISIBLE_DEVICES")
    return idx


def gpu_numa_node(device: int, sysfs: str = "/sys", env=os.environ):
    """NUMA node of HIP device `device` of this process, None when sysfs does not say"""
    addrs = gpu_pci_addresses(sysfs)
    vis = visible_indices(len(addrs), env)
    if device < 0 or device >= len(vis):
        return None
    text = _read(os.path.join(sysfs, "bus/pci/devices", addrs[vis[device]], "numa_node"))
    try:
        node = int(text) if text is not None else -1
    except ValueError:
        node = -1
    return node if node >= 0 else None


def node_cpus(node: int, sysfs: str = "/sys") -> list[int]:
    text = _read(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist"))
    return parse_cpulist(text) if text else []


def rank_cores(local_rank: int, local_world: int, allowed: list[int], sysfs: str = "/sys", env=os.environ, devices=None):
    """Cores for rank `local_rank` of `local_world` ranks on this host (rank r drives HIP device devices[r], default r).

    Ranks whose GPUs share a NUMA node split that node's allowed cores evenly in rank order; a rank whose GPU's node is unknown, or
    whose node has no allowed core, falls back to an even linear slice of the allowed cores.  Returns (cores, numa node or None, how)
    with how in {"numa", "linear"}."""
    allowed = sorted(allowed)
    devices = list(range(local_world)) if devices is None else list(devices)
    nodes = [gpu_numa_node(d, sysfs, env) for d in devices]
    aset = set(allowed)

    def numa_slice(r: int):
        """rank r's share of its GPU's node, None when the node is unknown, has no allowed core, or fewer cores than ranks"""
        node = nodes[r] if r < len(nodes) else None
        if node is None:
            return None
        cpus = [c for c in node_cpus(node, sysfs) if c in aset]
        peers = [q for q in range(local_world) if q < len(nodes) and nodes[q] == node]
        per = len(cpus) // len(peers) if peers else 0
        if per < 1:
            return None
        k = peers.index(r)
        return cpus[k * per:(k + 1) * per]

    mine = nodes[local_rank] if local_rank < len(nodes) else None
    got = numa_slice(local_rank)
    if got:
        return got, mine, "numa"
    # linear fallback: an even slice of the cores NO NUMA-placed rank of this host has claimed (a mixed host -- some GPUs with a known
    # node, some without -- must not put two ranks on the same cores), in the order of the ranks that fall back
    claimed = set()
    fallers = []
    for r in range(local_world):
        sl = numa_slice(r)
        if sl:
            claimed.update(sl)
        else:
            fallers.append(r)
    free = [c for c in allowed if c not in claimed]
    if not free:                       # every allowed core is some NUMA-placed rank's: overlap is unavoidable, take the plain even slice
        per = max(1, len(allowed) // max(1, local_world))
        return (allowed[local_rank * per:(local_rank + 1) * per] or allowed), mine, "linear"
    per = max(1, len(free) // max(1, len(fallers)))
    k = fallers.index(local_rank) if local_rank in fallers else 0
    return (free[k * per:(k + 1) * per] or free), mine, "linear"
