"""Host plumbing of a rank (diffsheg_amd/hostenv.py) against a fake sysfs tree: NUMA-local affinity planning and the clock / power
sampler.  No GPU, no real /sys."""
import os
import time

from diffsheg_amd import hostenv


def _fake_tree(tmp_path, n_cards=4):
    drm = tmp_path / "drm"; nodes = tmp_path / "node"
    for i in range(n_cards):
        dev = tmp_path / "pci" / f"0000:{i + 1:02x}:00.0"
        hw = dev / "hwmon" / f"hwmon{i}"
        hw.mkdir(parents=True)
        (dev / "vendor").write_text("0x1002\n")
        (dev / "numa_node").write_text(f"{i // 2}\n")
        (hw / "freq1_input").write_text(str((1800 + 10 * i) * 1000000) + "\n")
        (hw / "power1_average").write_text(str((1000 + i) * 1000000) + "\n")
        card = drm / f"card{i}"
        card.mkdir(parents=True)
        os.symlink(dev, card / "device")
    for n in range(2):
        d = nodes / f"node{n}"; d.mkdir(parents=True)
        (d / "cpulist").write_text(f"{n * 8}-{n * 8 + 3},{n * 8 + 4}-{n * 8 + 7}\n")
    return str(drm), str(nodes)


def test_cpulist_parsing_and_affinity_plan():
    assert hostenv._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    cpus = list(range(16))
    assert hostenv.plan_affinity(cpus, [0, 1, 2, 3], 0) == [0, 1, 2, 3]
    assert hostenv.plan_affinity(cpus, [4, 5, 6, 7], 7) == [12, 13, 14, 15]
    assert hostenv.plan_affinity(list(range(5)), [0, 1], 1) == [2, 3, 4]          # the last rank takes the remainder
    assert hostenv.plan_affinity([3], [0, 1, 2], 1) == [3]                        # fewer cores than ranks: share


def test_numa_lookup_on_a_fake_tree(tmp_path):
    drm, nodes = _fake_tree(tmp_path)
    assert len(hostenv.amdgpu_cards(drm)) == 4
    assert hostenv.numa_cpus_of_gpu(0, drm, nodes) == list(range(0, 8))
    assert hostenv.numa_cpus_of_gpu(3, drm, nodes) == list(range(8, 16))
    assert hostenv.numa_cpus_of_gpu(9, drm, nodes) is None


def test_pinning_reports_and_never_raises(tmp_path, monkeypatch):
    drm, nodes = _fake_tree(tmp_path)
    before = os.sched_getaffinity(0)
    try:
        info = hostenv.pin_to_local_numa(1, 4, drm, nodes)        # GPU 1 hangs off node 0 together with GPU 0: second half of cores 0..7
        if info["pinned"]:
            assert set(os.sched_getaffinity(0)) <= {4, 5, 6, 7} and info["ranks_on_node"] == 2
        else:
            assert "why" in info                                   # this container's cpuset may not contain those cores
        os.sched_setaffinity(0, before)
        ids = ["0000:01:00.0", "0000:04:00.0"]                      # a container that drives cards 0 and 3 of a host showing four
        info = hostenv.pin_to_local_numa(1, 2, drm, nodes, pci_ids=ids)   # device 1 = card 3 = node 1, alone on it
        if info["pinned"]:
            assert set(os.sched_getaffinity(0)) <= set(range(8, 16)) and info["ranks_on_node"] == 1
        os.sched_setaffinity(0, before)
        assert hostenv.pin_to_local_numa(0, 1, drm, nodes, pci_ids=[None])["pinned"] is False
    finally:
        os.sched_setaffinity(0, before)
    monkeypatch.setenv("DSH_PIN", "0")
    assert hostenv.pin_to_local_numa(0, 1, drm, nodes) == {"pinned": False, "why": "DSH_PIN=0"}
    monkeypatch.delenv("DSH_PIN")
    assert hostenv.pin_to_local_numa(0, 1, str(tmp_path / "nothing"), nodes)["pinned"] is False


def test_telemetry_sampler_on_a_fake_tree(tmp_path):
    drm, _ = _fake_tree(tmp_path)
    # several cards on the host and no PCI address: unknown, never a guess
    assert not hostenv.GpuTelemetry(2, sysfs=drm).available
    assert hostenv.card_of_device("0000:03:00.0", 0, drm).endswith("0000:03:00.0") and hostenv.card_of_device("03:00.0", 5, drm) is not None
    assert hostenv.card_of_device("0000:7f:00.0", 0, drm) is None
    t = hostenv.GpuTelemetry(0, period_s=0.005, sysfs=drm, pci_bus_id="0000:03:00.0")
    assert t.available
    t.start(); time.sleep(0.06); t.stop()
    s = t.summary()
    assert s["samples"] >= 3 and abs(s["clock_mhz_mean"] - 1820.0) < 1e-6 and abs(s["power_w_mean"] - 1002.0) < 1e-6
    none = hostenv.GpuTelemetry(0, sysfs=str(tmp_path / "nothing"))
    assert not none.available and none.summary()["clock_mhz_mean"] is None
