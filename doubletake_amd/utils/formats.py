"""On-disk formats either side of the hot path, so that outputs of this
package can be diffed against files written by the reference (and its published result tarballs).

  * fused volume  : ``.npz`` with the five arrays of ``TSDF.save_tsdf`` (reference tools/tsdf.py:267-275)
                    -- written/read by ``doubletake_amd.tools.tsdf.TSDF.save_tsdf / from_file``;
                    ``tsdf_npz_manifest`` describes a file for comparisons.
  * depth cache   : one ``<frame_id>.pickle`` per keyframe holding the predicted depth, mask, optional
                    confidence, intrinsics, pose and ids (reference utils/generic_utils.py:304-352;
                    read back by tools/partial_fuser.py:22-33).
  * score sheet   : JSON written by ``ResultsAverager.output_json`` (reference utils/metrics_utils.py:200-239).
  * mesh          : binary little-endian PLY (vertices float32 x/y/z, faces uchar+3*int32).

Pure host code: no GPU work happens here.
"""
from __future__ import annotations

import json
import os
import pickle
from collections import OrderedDict

import numpy as np

TSDF_NPZ_KEYS = ("tsdf_values", "tsdf_weights", "origin", "voxel_coords_3hwd", "voxel_size")
DEPTH_CACHE_KEYS = ("depth_pred_s0_b1hw", "overall_mask_bhw", "K_full_depth_b44", "K_s0_b44", "cam_T_world_b44",
                    "frame_id", "src_ids")


def tsdf_npz_manifest(path):
    """{key: (dtype string, shape)} of a saved volume."""
    with np.load(path) as data:
        return {k: (str(data[k].dtype), tuple(data[k].shape)) for k in data.files}


# ---- depth cache ----------------------------------------------------------------------------------
def write_depth_cache(output_path, outputs, cur_data, src_data, batch_ind=0, batch_size=1):
    """One pickle per batch element, named by its frame id string; every tensor keeps a leading
    batch axis of 1 (generic_utils.py:304-352).  ``cv_confidence_b1hw`` is stored when present."""
    os.makedirs(output_path, exist_ok=True)
    written = []
    n = outputs["depth_pred_s0_b1hw"].shape[0]
    for e in range(n):
        if "frame_id_string" in cur_data:
            frame_id = cur_data["frame_id_string"][e]
        else:
            frame_id = f"{batch_ind * batch_size + e:6d}"
        rec = {name: outputs[name][e].unsqueeze(0) for name in ("depth_pred_s0_b1hw", "overall_mask_bhw")}
        if "cv_confidence_b1hw" in outputs:
            rec["cv_confidence_b1hw"] = outputs["cv_confidence_b1hw"][e].unsqueeze(0)
        for name in ("K_full_depth_b44", "K_s0_b44", "cam_T_world_b44"):
            rec[name] = cur_data[name][e].unsqueeze(0)
        rec["frame_id"] = cur_data["frame_id_string"][e] if "frame_id_string" in cur_data else frame_id
        rec["src_ids"] = [ids[e] for ids in src_data["frame_id_string"]]
        path = os.path.join(output_path, f"{frame_id}.pickle")
        with open(path, "wb") as fh:
            pickle.dump(rec, fh)
        written.append(path)
    return written


def read_depth_cache(cached_depth_path):
    """OrderedDict {int(frame id): record}, ascending (what PartialFuser builds, partial_fuser.py:22-40)."""
    found = {}
    for name in os.listdir(cached_depth_path):
        if name.endswith(".pickle"):
            with open(os.path.join(cached_depth_path, name), "rb") as fh:
                found[int(name.split(".")[0])] = pickle.load(fh)
    return OrderedDict((k, found[k]) for k in sorted(found))


# ---- score sheet ----------------------------------------------------------------------------------
def write_scores_json(filepath, exp_name, metrics_name, scores):
    """scores: ordered {metric: number}.  Same keys, float formatting and indent as the reference."""
    names_row, values_row = "", ""
    out_scores = {}
    for k, v in scores.items():
        names_row += f"{k:8} "
        cell = f"{v:.4f},"
        values_row += f"{cell:8} "
        out_scores[k] = float(v)
    doc = {"exp_name": exp_name, "metrics_type": metrics_name, "scores": out_scores,
           "metrics_string": names_row, "scores_string": values_row}
    with open(filepath, "w") as fh:
        json.dump(doc, fh, indent=4)
    return doc


def read_scores_json(filepath):
    with open(filepath) as fh:
        return json.load(fh)


# ---- PLY ------------------------------------------------------------------------------------------
def write_ply(path, verts, faces, comment=None):
    v = np.ascontiguousarray(np.asarray(verts, dtype="<f4").reshape(-1, 3))
    f = np.ascontiguousarray(np.asarray(faces, dtype="<i4").reshape(-1, 3))
    note = f"comment {comment}\n" if comment else ""
    header = (f"ply\nformat binary_little_endian 1.0\n{note}element vertex {len(v)}\nproperty float x\nproperty float y\n"
              f"property float z\nelement face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n")
    rec = np.empty(len(f), dtype=[("n", "u1"), ("idx", "<i4", 3)])
    rec["n"] = 3
    rec["idx"] = f
    with open(path, "wb") as fh:
        fh.write(header.encode())
        fh.write(v.tobytes())
        fh.write(rec.tobytes())


def read_ply(path):
    """(verts [N,3] float32, faces [M,3] int32) of a binary-little-endian or ascii triangle PLY whose
    vertex element starts with float x, y, z."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elems, cur = None, [], None
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elems.append(cur)
            elif tok[0] == "property":
                cur["props"].append(tok[1:])
            elif tok[0] == "end_header":
                break
        sizes = {"char": 1, "uchar": 1, "int8": 1, "uint8": 1, "short": 2, "ushort": 2, "int16": 2, "uint16": 2,
                 "int": 4, "uint": 4, "int32": 4, "uint32": 4, "float": 4, "float32": 4, "double": 8, "float64": 8}
        codes = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2", "int16": "<i2",
                 "uint16": "<u2", "int": "<i4", "uint": "<u4", "int32": "<i4", "uint32": "<u4", "float": "<f4",
                 "float32": "<f4", "double": "<f8", "float64": "<f8"}
        verts = np.zeros((0, 3), np.float32)
        faces = np.zeros((0, 3), np.int32)
        for el in elems:
            if fmt == "ascii":
                rows = [fh.readline().split() for _ in range(el["count"])]
                if el["name"] == "vertex":
                    verts = np.array([[float(x) for x in r[:3]] for r in rows], dtype=np.float32).reshape(-1, 3)
                elif el["name"] == "face":
                    faces = np.array([[int(x) for x in r[1:4]] for r in rows], dtype=np.int32).reshape(-1, 3)
                continue
            if fmt != "binary_little_endian":
                raise NotImplementedError(f"{path}: PLY format {fmt}")
            if el["name"] == "vertex":
                dt = np.dtype([(p[1], codes[p[0]]) for p in el["props"]])
                raw = np.frombuffer(fh.read(dt.itemsize * el["count"]), dtype=dt)
                verts = np.stack([raw["x"], raw["y"], raw["z"]], axis=1).astype(np.float32)
            elif el["name"] == "face":
                prop = el["props"][0]
                if prop[0] != "list":
                    raise NotImplementedError(f"{path}: face element without a list property")
                cnt_sz, idx_code = sizes[prop[1]], codes[prop[2]]
                out = np.empty((el["count"], 3), np.int32)
                for i in range(el["count"]):
                    n = int.from_bytes(fh.read(cnt_sz), "little")
                    idx = np.frombuffer(fh.read(n * sizes[prop[2]]), dtype=idx_code)
                    if n != 3:
                        raise NotImplementedError(f"{path}: non-triangle face")
                    out[i] = idx
                faces = out
            else:
                width = sum(sizes[p[0]] for p in el["props"])
                fh.read(width * el["count"])
    return verts, faces
