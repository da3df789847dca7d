"""PLY checkpoint interchange with the reference.

`save_ply` writes exactly what /root/reference/scene/gaussian_model.py:761-804 hands to plyfile: one `vertex`
element, every property a little-endian float32, in the order of construct_list_of_attributes (:696-725):
    x y z trbf_center trbf_scale nx ny nz f_dc_* f_rest_* f_t_* motion_* opacity scale_* rot_* omega_* zeta_*
    control_{x,y,z}_k (k = 0..11) current_control_num
plus the decoder's state_dict next to it as <name>.pt.  `load_ply` reads the properties back by NAME, as
load_ply (:934-1040) does, so files written by the reference load here and vice versa.  No plyfile dependency:
the binary_little_endian 1.0 container is a text header + packed rows.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch


def attribute_names(pc) -> List[str]:
    names = ["x", "y", "z", "trbf_center", "trbf_scale", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(pc._features_dc.shape[1])]
    fr = pc._features_rest
    names += [f"f_rest_{i}" for i in range(fr.shape[1] * fr.shape[2])]
    names += [f"f_t_{i}" for i in range(pc._features_t.shape[1])]
    names += [f"motion_{i}" for i in range(pc._motion.shape[1])]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(pc._scaling.shape[1])]
    names += [f"rot_{i}" for i in range(pc._rotation.shape[1])]
    names += [f"omega_{i}" for i in range(pc._omega.shape[1])]
    names += [f"zeta_{i}" for i in range(pc._zeta.shape[1])]
    for k in range(pc.control_xyz.shape[1]):
        names += [f"control_x_{k}", f"control_y_{k}", f"control_z_{k}"]
    names.append("current_control_num")
    return names


def attribute_rows(pc) -> np.ndarray:
    """[N, len(attribute_names)] float32, the row matrix of save_ply (:789)."""
    def a(t):
        return t.detach().reshape(t.shape[0], -1).to(torch.float32).cpu().numpy()

    xyz = a(pc._xyz)
    cols = [xyz, a(pc._trbf_center), a(pc._trbf_scale), np.zeros_like(xyz), a(pc._features_dc),
            a(pc._features_rest.detach().transpose(1, 2)), a(pc._features_t), a(pc._motion),
            a(pc._opacity), a(pc._scaling), a(pc._rotation), a(pc._omega), a(pc._zeta), a(pc.control_xyz),
            a(pc.current_control_num)]
    return np.concatenate(cols, axis=1).astype("<f4")


def write_ply(path: str, names: List[str], rows: np.ndarray) -> None:
    rows = np.ascontiguousarray(rows, dtype="<f4")
    assert rows.ndim == 2 and rows.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {rows.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rows.tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}


def read_ply(path: str) -> Tuple[List[str], np.ndarray]:
    """First element of a binary_little_endian / ascii PLY -> (property names, [N, P] float64 matrix)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, seen_element = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header not terminated")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_element:  # only the first element is read; skip the rest of the header
                    while f.readline().strip() != b"end_header":
                        pass
                    break
                seen_element, count = True, int(tok[2])
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [n for n, _ in props]
        if fmt == "binary_little_endian":
            dt = np.dtype([(n, t) for n, t in props])
            data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            rows = np.stack([data[n].astype(np.float64) for n in names], 1) if names else np.zeros((count, 0))
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2).astype(np.float64)
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return names, rows


def save_ply(pc, path: str) -> None:
    write_ply(path, attribute_names(pc), attribute_rows(pc))
    torch.save(pc.rgbdecoder.state_dict(), path.replace(".ply", ".pt"))


def load_ply(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """-> (params, dynamic) dictionaries for GaussianParams / TrainableGaussians (the constructor arguments)."""
    names, rows = read_ply(path)
    col = {n: i for i, n in enumerate(names)}

    def take(prefix, sort=True):
        ks = [n for n in names if n.startswith(prefix)]
        if sort:
            ks = sorted(ks, key=lambda s: int(s.split("_")[-1]))
        return torch.tensor(rows[:, [col[k] for k in ks]], dtype=torch.float32) if ks else \
            torch.zeros(rows.shape[0], 0)

    one = lambda n: torch.tensor(rows[:, [col[n]]], dtype=torch.float32)  # noqa: E731
    n_ctrl = len([n for n in names if n.startswith("control_x_")])
    ctrl = torch.stack([torch.tensor(rows[:, [col[f"control_{a}_{k}"] for a in "xyz"]], dtype=torch.float32)
                        for k in range(n_ctrl)], 1)
    params = {"xyz": torch.tensor(rows[:, [col["x"], col["y"], col["z"]]], dtype=torch.float32),
              "scaling": take("scale_"), "rotation": take("rot_"), "opacity": one("opacity"),
              "features_dc": take("f_dc_"), "features_t": take("f_t_")}
    f_rest = take("f_rest_")
    dynamic = {"omega": take("omega_"), "zeta": take("zeta_"), "trbf_center": one("trbf_center"),
               "trbf_scale": one("trbf_scale"), "motion": take("motion_"), "control_xyz": ctrl,
               "current_control_num": torch.tensor(rows[:, [col["current_control_num"]]]).round().to(torch.int64),
               "f_rest": f_rest.reshape(rows.shape[0], 3, -1).transpose(1, 2) if f_rest.shape[1]
               else torch.zeros(rows.shape[0], 0, 3)}
    return params, dynamic
