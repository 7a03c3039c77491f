"""extract_vectors_device on the 16-size JPEG list (64 per size) with different numbers of loader workers: on a box whose
cgroup grants fewer CPUs than it shows (cpu.max), too many busy workers get the whole process group throttled."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bench_extract import LIST_SHAPES, _write_jpegs
from mdir_amd.datasets import initialize_transforms
from mdir_amd.network import CirNetwork, SingleNetwork
from mdir_amd.networks import extract_vectors_device, init_network

dev = torch.device("cuda:0")
torch.manual_seed(3)
model = init_network({"architecture": "resnet101", "pooling": "gem", "whitening": False, "pretrained": False})
D = model.meta["outputdim"]
model.meta["in_channels"], model.meta["out_channels"] = 3, D
rng = np.random.default_rng(2)
qm, _ = np.linalg.qr(rng.standard_normal((D, D)))
wh = {"P": (qm * rng.uniform(0.5, 2.0, (1, D))).T.copy(), "m": rng.normal(0, 0.01, (D, 1))}
mp = {"architecture": "cirnet", "cir_architecture": "resnet101", "local_whitening": False, "pooling": "gem", "regional": False, "whitening": False, "pretrained": False}
rt = {"wrappers": {"train": "", "eval": {"0_cirwhiten": {"whitening": wh, "dimensions": None}, "1_cirmultiscale": {"scales": True}}}, "data": {"transforms": "pil2np | totensor | normalize"}}
net = CirNetwork(model.to(dev), SingleNetwork.NetworkParams(mp, rt), dev, frozen=True).eval()
tr = initialize_transforms("pil2np | totensor | normalize", net.network_params.runtime["data"]["mean_std"])
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "os.cpu_count():", os.cpu_count())
except OSError:
    pass
with tempfile.TemporaryDirectory() as folder:
    files = _write_jpegs(folder, LIST_SHAPES, 8)
    paths = [f for r in range(8) for f in files]
    with torch.no_grad():
        extract_vectors_device(net, paths[:256], 1024, tr, device=dev, num_workers=4, print_freq=10 ** 9)      # MIOpen warm-up
        for spec in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,6,8").split(","):
            os.environ["MDIR_AMD_LOADER"] = "threads" if spec.startswith("t") else "processes"
            wk = int(spec.lstrip("t"))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            extract_vectors_device(net, paths, 1024, tr, device=dev, num_workers=wk, print_freq=10 ** 9)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(os.environ["MDIR_AMD_LOADER"], "workers %d: %.2f ms per image, %.1f descriptors/s (list of %d, graph captures included)" % (wk, 1e3 * dt / len(paths), len(paths) / dt, len(paths)), flush=True)
