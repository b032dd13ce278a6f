"""CPU tests of the batch / checkpoint helpers either side of the training path: the train/validation split of
vae/training/train.py:156-159, the state_dict file of train.py:360-367 and the image grid writer.  No GPU."""
import numpy as np
import pytest
import torch
from torch.utils.data import TensorDataset, random_split

from vae_amd import data


@pytest.mark.parametrize("n,p", [(10, 0.9), (50000, 0.9), (7, 0.5), (3, 1.0), (5, 0.0)])
def test_split_matches_random_split(n, p):
    tr, va = data.split_indices(n, p, torch.Generator().manual_seed(1337))
    assert tr.numel() == int(n * p) and va.numel() == n - int(n * p)
    assert sorted(torch.cat([tr, va]).tolist()) == list(range(n))
    # the same indices torch's random_split hands to its two Subsets under the same generator
    ds = TensorDataset(torch.arange(n))
    a, b = random_split(ds, [int(n * p), n - int(n * p)], generator=torch.Generator().manual_seed(1337))
    assert list(a.indices) == tr.tolist() and list(b.indices) == va.tolist()


def test_split_rejects_bad_arguments():
    with pytest.raises(ValueError):
        data.split_indices(0, 0.9)
    with pytest.raises(ValueError):
        data.split_indices(10, 1.5)


def test_checkpoint_round_trip_is_a_plain_state_dict(tmp_path):
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    path = str(tmp_path / "run" / "best-model.pth")
    data.save_checkpoint(m, path)
    raw = torch.load(path, map_location="cpu", weights_only=True)      # what the reference's loader reads (train.py:187)
    assert list(raw.keys()) == list(m.state_dict().keys())
    assert all(v.device.type == "cpu" for v in raw.values())
    m2 = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    res = data.load_checkpoint(m2, path)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v)
    torch.save([1, 2, 3], str(tmp_path / "junk.pth"))
    with pytest.raises(TypeError):
        data.load_checkpoint(m2, str(tmp_path / "junk.pth"))


def test_device_feed_validates_on_the_host():
    with pytest.raises(TypeError):
        data.DeviceFeed(np.zeros((4, 8, 8, 3), np.float32), 2, device="cpu")
    with pytest.raises(ValueError):
        data.DeviceFeed(np.zeros((4, 8, 8, 2), np.uint8), 2, device="cpu")
    with pytest.raises(IndexError):
        data.DeviceFeed(np.zeros((4, 8, 8, 3), np.uint8), 2, indices=torch.tensor([0, 4]), device="cpu")
    with pytest.raises(ValueError):
        data.DeviceFeed(np.zeros((4, 8, 8, 3), np.uint8), 0, device="cpu")
    feed = data.DeviceFeed(np.zeros((5, 8, 8), np.uint8), 2, device="cpu")      # construction touches no kernel
    assert len(feed) == 3 and (feed.c, feed.h, feed.w) == (1, 8, 8)


def test_image_grid_layout(tmp_path):
    imgs = torch.zeros(5, 3, 4, 6)
    for i in range(5):
        imgs[i, i % 3] = 1.0
    path = str(tmp_path / "grid.ppm")
    data.save_image_grid(imgs, path, nrow=3, padding=1)
    raw = open(path, "rb").read()
    header, body = raw.split(b"\n255\n", 1)
    assert header == b"P6\n22 11"                     # 3 columns x (6 + 1) + 1, 2 rows x (4 + 1) + 1
    arr = np.frombuffer(body, np.uint8).reshape(11, 22, 3)
    assert (arr[0] == 0).all() and (arr[:, 0] == 0).all()          # padding
    assert (arr[1:5, 1:7] == [255, 0, 0]).all()                    # image 0 is red
    assert (arr[6:10, 8:14] == [0, 255, 0]).all()                  # image 4 (row 1, column 1) is green
    assert (arr[6:10, 15:21] == 0).all()                           # the empty cell stays black
    data.save_image_grid(imgs[:, :1], str(tmp_path / "grey.png"))  # one channel, through PIL


# ---------------------------------------------------------------------- CIFAR-10 batch files (train.py:266-274)
def _write_cifar_like(root, n_per_file=7, seed=0):
    """files in the layout of the CIFAR-10 python version, written the way the originals were: a pickled dict with a
    uint8 [n, 3072] array of CHW planes and a list of labels"""
    import os
    import pickle
    import numpy as np
    base = os.path.join(root, "cifar-10-batches-py")
    os.makedirs(base)
    rng = np.random.default_rng(seed)
    all_data, all_labels = {}, {}
    for name in [f"data_batch_{i}" for i in range(1, 6)] + ["test_batch"]:
        data = rng.integers(0, 256, size=(n_per_file, 3072), dtype=np.uint8)
        labels = [int(v) for v in rng.integers(0, 10, size=n_per_file)]
        with open(os.path.join(base, name), "wb") as f:
            pickle.dump({"batch_label": name, "labels": labels, "data": data, "filenames": [f"{name}_{i}.png" for i in range(n_per_file)]},
                        f, protocol=2)
        all_data[name], all_labels[name] = data, labels
    return all_data, all_labels


def test_load_cifar10_reads_the_batch_files_in_dataset_order(tmp_path):
    import numpy as np
    from vae_amd.data import load_cifar10
    data, labels = _write_cifar_like(str(tmp_path))
    images, lab = load_cifar10(str(tmp_path), train=True)
    assert images.shape == (35, 32, 32, 3) and images.dtype == np.uint8 and lab.shape == (35,) and lab.dtype == np.int64
    want = np.concatenate([data[f"data_batch_{i}"] for i in range(1, 6)]).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    assert np.array_equal(images, want)                       # CHW planes -> HWC pixels, files in order
    assert lab.tolist() == sum((labels[f"data_batch_{i}"] for i in range(1, 6)), [])
    test_images, test_lab = load_cifar10(str(tmp_path), train=False)
    assert np.array_equal(test_images, data["test_batch"].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
    assert test_lab.tolist() == labels["test_batch"]


def test_load_cifar10_executes_nothing_from_the_file(tmp_path):
    """a pickle that names any other global (here: os.system) is refused before anything is called"""
    import os
    import pickle
    import pytest
    from vae_amd.data import load_cifar10
    _write_cifar_like(str(tmp_path))
    marker = os.path.join(str(tmp_path), "executed")

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    with open(os.path.join(str(tmp_path), "cifar-10-batches-py", "data_batch_3"), "wb") as f:
        pickle.dump({"data": Evil(), "labels": []}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError, match="refusing to load global"):
        load_cifar10(str(tmp_path), train=True)
    assert not os.path.exists(marker)
    with pytest.raises(FileNotFoundError):
        load_cifar10(os.path.join(str(tmp_path), "nowhere"))
