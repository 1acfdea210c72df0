#!/usr/bin/env python3
"""Which sysfs files of the GPU's PCI device an ordinary user can read for power / clocks (GPU box): what bench.py's power sample rests on."""
import glob
import os

import torch

p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
base = "/sys/bus/pci/devices/" + bdf
print("bdf", bdf, "exists", os.path.exists(base))
for h in glob.glob(base + "/hwmon/hwmon*"):
    for f in sorted(os.listdir(h)):
        fp = os.path.join(h, f)
        if os.path.isfile(fp) and (f.startswith("power") or f.startswith("freq") or f.startswith("temp1") or f == "name"):
            try:
                print(os.path.basename(h), f, open(fp).read().strip()[:60])
            except Exception as e:  # noqa: BLE001
                print(os.path.basename(h), f, "ERR", e)
for f in ("pp_dpm_sclk", "gpu_busy_percent"):
    try:
        print(f, open(base + "/" + f).read().strip().replace("\n", " | ")[:200])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
