"""Host-side plumbing around a rank: NUMA-local CPU affinity and a clock / power sampler for the GPU it drives.

Neither touches the data path.  One process per GPU issues ~12 000 kernel launches per 950-clip step (reference: one process per GPU
under mp.spawn, runner.py:80-122); on an 8-GPU node the eight launch threads should sit on cores of the socket their GPU hangs off,
and the benchmark line should say at which shader clock and socket power its step ran (the part clocks to its power budget:
MI355X_MICROARCH.md, "DVFS give-back").  Everything here reads sysfs only (amdgpu hwmon / NUMA topology) and degrades to "unknown"
— None values — when a file is missing: no dependency on rocm-smi or amdsmi being importable.
"""
from __future__ import annotations

import glob
import os
import threading
import time
from typing import Dict, List, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _parse_cpulist(txt: str) -> List[int]:
    cpus: List[int] = []
    for part in txt.split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def amdgpu_cards(sysfs: str = "/sys/class/drm") -> List[str]:
    """device directories of the amdgpu cards that expose a hwmon node, in PCI-address order (= HIP's default device order)"""
    cards = []
    for d in glob.glob(os.path.join(sysfs, "card[0-9]*", "device")):
        if glob.glob(os.path.join(d, "hwmon", "hwmon*")) and (_read(os.path.join(d, "vendor")) or "").lower() == "0x1002":
            cards.append(os.path.realpath(d))
    return sorted(set(cards))


def numa_cpus_of_card(card: Optional[str], nodes: str = "/sys/devices/system/node") -> Optional[List[int]]:
    if card is None:
        return None
    node = _read(os.path.join(card, "numa_node"))
    if node is None or int(node) < 0:
        return None
    cl = _read(os.path.join(nodes, f"node{int(node)}", "cpulist"))
    return _parse_cpulist(cl) if cl else None


def numa_cpus_of_gpu(local_rank: int, sysfs: str = "/sys/class/drm", nodes: str = "/sys/devices/system/node",
                     pci_bus_id: Optional[str] = None) -> Optional[List[int]]:
    """cores of the NUMA node GPU `local_rank` hangs off.  pci_bus_id given: the card is found by address (card_of_device); else the
    host's cards are indexed in PCI order, which is right only when HIP enumerates all of them"""
    if pci_bus_id:
        return numa_cpus_of_card(card_of_device(pci_bus_id, local_rank, sysfs), nodes)
    cards = amdgpu_cards(sysfs)
    return numa_cpus_of_card(cards[local_rank], nodes) if local_rank < len(cards) else None


def plan_affinity(cpus: List[int], ranks_on_node: List[int], local_rank: int) -> List[int]:
    """the slice of a NUMA node's CPU list for one of the ranks whose GPUs hang off that node (contiguous, equal shares)"""
    ranks = sorted(ranks_on_node)
    i, n = ranks.index(local_rank), len(ranks)
    per = max(1, len(cpus) // n)
    mine = cpus[i * per:(i + 1) * per] if i < n - 1 else cpus[i * per:]
    return mine or cpus


def pin_to_local_numa(local_rank: int, local_world: int, sysfs: str = "/sys/class/drm", nodes: str = "/sys/devices/system/node",
                      pci_ids: Optional[List[Optional[str]]] = None) -> Dict:
    """os.sched_setaffinity for this process: the cores of its GPU's NUMA node, shared evenly with the other local ranks on that node.
    pci_ids[r] = PCI address of local device r (GpuTelemetry.pci_bus_id_of) — containers show every GPU of the host in sysfs, so the
    rank index alone does not identify the card.  Returns what was done ({"pinned": False, "why": ...} when the topology is not
    visible).  DSH_PIN=0 disables it."""
    if os.environ.get("DSH_PIN") == "0":
        return {"pinned": False, "why": "DSH_PIN=0"}
    if not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "why": "no sched_setaffinity on this platform"}
    pid = lambda r: (pci_ids[r] if pci_ids and r < len(pci_ids) else None)      # noqa: E731
    if pci_ids is not None and pid(local_rank) is None:
        return {"pinned": False, "why": "PCI address of this rank's device unknown"}
    mine = numa_cpus_of_gpu(local_rank, sysfs, nodes, pid(local_rank))
    if not mine:
        return {"pinned": False, "why": "GPU NUMA node not visible in sysfs"}
    same = [r for r in range(local_world) if (pci_ids is None or pid(r)) and numa_cpus_of_gpu(r, sysfs, nodes, pid(r)) == mine]
    if local_rank not in same:
        same.append(local_rank)
    allowed = set(os.sched_getaffinity(0))
    want = [c for c in plan_affinity(mine, same, local_rank) if c in allowed]
    if not want:
        return {"pinned": False, "why": "NUMA-local cores are outside this process's cpuset"}
    os.sched_setaffinity(0, want)
    return {"pinned": True, "cpus": len(want), "first_cpu": want[0], "last_cpu": want[-1], "ranks_on_node": len(same)}


def card_of_device(pci_bus_id: Optional[str], local_rank: int, sysfs: str = "/sys/class/drm") -> Optional[str]:
    """sysfs device directory of the GPU a process drives.  A container usually sees the hwmon nodes of EVERY GPU of the host while
    HIP enumerates only the ones assigned to it, so the index alone is not enough: match the PCI address ("0000:c1:00.0", any case;
    a bare "c1:00.0" matches domain 0).  Without an address the index is trusted only when the host shows exactly as many cards as
    local ranks could need (pci_bus_id = None and several cards -> None: unknown, never a guess)."""
    cards = amdgpu_cards(sysfs)
    if pci_bus_id:
        want = pci_bus_id.strip().lower()
        if want.count(":") == 1:
            want = "0000:" + want
        for c in cards:
            if os.path.basename(c).lower() == want:
                return c
        return None
    return cards[local_rank] if len(cards) == 1 and local_rank == 0 else None


class GpuTelemetry:
    """Background sampler of one GPU's shader clock and socket power (amdgpu hwmon: freq1_input in Hz, power1_average / power1_input
    in microwatts).  start() / stop() bracket a region; summary() gives the means over the samples taken while it ran."""

    def __init__(self, local_rank: int = 0, period_s: float = 0.02, sysfs: str = "/sys/class/drm", pci_bus_id: Optional[str] = None,
                 by_index: bool = False):
        self.period = period_s
        self.freq_path = self.power_path = None
        cards = amdgpu_cards(sysfs)
        card = cards[local_rank] if (by_index and local_rank < len(cards)) else card_of_device(pci_bus_id, local_rank, sysfs)
        self.card = card
        if card is not None:
            for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
                f = os.path.join(hw, "freq1_input")
                if self.freq_path is None and _read(f) is not None:
                    self.freq_path = f
                for name in ("power1_average", "power1_input"):
                    pth = os.path.join(hw, name)
                    if self.power_path is None and _read(pth) not in (None, "", "0"):
                        self.power_path = pth
        self._mhz: List[float] = []
        self._watt: List[float] = []
        self._stop = threading.Event()
        self._thr: Optional[threading.Thread] = None

    @property
    def available(self) -> bool:
        return self.freq_path is not None or self.power_path is not None

    def _run(self):
        while not self._stop.is_set():
            if self.freq_path:
                v = _read(self.freq_path)
                if v and v.isdigit():
                    self._mhz.append(int(v) / 1e6)
            if self.power_path:
                v = _read(self.power_path)
                if v and v.isdigit():
                    self._watt.append(int(v) / 1e6)
            self._stop.wait(self.period)

    def start(self):
        if not self.available or self._thr is not None:
            return
        self._stop.clear()
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr is None:
            return
        self._stop.set()
        self._thr.join(timeout=2.0)
        self._thr = None

    @staticmethod
    def pci_bus_id_of(device_index: int) -> Optional[str]:
        """PCI address of a visible HIP device ("0000:c1:00.0") from torch's device properties, or None"""
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            dom, bus, dv = getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
            if bus is None or dv is None:
                return None
            return "%04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(dv))
        except Exception:                                           # noqa: BLE001
            return None

    def summary(self) -> Dict:
        mean = lambda xs: (sum(xs) / len(xs)) if xs else None       # noqa: E731
        return {"clock_mhz_mean": mean(self._mhz), "clock_mhz_min": min(self._mhz) if self._mhz else None,
                "clock_mhz_max": max(self._mhz) if self._mhz else None, "power_w_mean": mean(self._watt),
                "power_w_max": max(self._watt) if self._watt else None, "samples": max(len(self._mhz), len(self._watt)),
                "source": "amdgpu hwmon (freq1_input = sclk, power1 = socket power), sampled every %d ms during the timed region" % int(self.period * 1e3)
                if self.available else "unavailable: no amdgpu hwmon node matched this device",
                "card": os.path.basename(self.card) if self.card else None}
