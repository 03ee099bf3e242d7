"""Volume ingest, MI355X-native: SURVEY.md 8(f) row f4.

Mirror of the reference's `utils.preprocess` (src/utils/preprocess.py) for the CT-CLIP pipeline.  The reference turns
a raw NIfTI scan into the model's `[1, 240, 480, 480]` input through five full-size host tensors (HU rescale, permute,
`F.interpolate(trilinear)`, clamp / 1000, centre crop-or-pad) and ships the f32 result to the device.  Here the raw
voxels go to the device once (int16 or float32) and `process_volume` produces the bf16 (or f32) model input with ONE
kernel, `ctclip_ingest_volume` (csrc/ingest.hip).

`process_file` keeps the reference's signature and metadata handling (:84-121); reading the NIfTI needs `nibabel`,
exactly as in the reference."""
from __future__ import annotations

import torch

from ctclip_hip.lib import hip

TARGET_SPACING = (1.5, 0.75, 0.75)          # (z, x, y)  reference :130
TARGET_SHAPE = (480, 480, 240)              # (H, W, D)  reference :143


def read_nii_data(file_path):
    """reference :8-18"""
    try:
        import nibabel as nib
    except ImportError as e:                 # pragma: no cover - nibabel is not part of this image
        raise ImportError("reading NIfTI files needs nibabel (as the reference does)") from e
    try:
        return nib.load(file_path).get_fdata()
    except Exception as e:
        print(f"Error reading file {file_path}: {e}")
        return None


def resampled_shape(shape_dhw, current_spacing, target_spacing=TARGET_SPACING):
    """new_shape of reference `resize_array` (:33-35): int(size * current / target) per axis, in double precision."""
    return tuple(int(shape_dhw[i] * (current_spacing[i] / target_spacing[i])) for i in range(3))


def process_volume(raw_hwd, slope, intercept, xy_spacing, z_spacing, *, target_shape=TARGET_SHAPE,
                   target_spacing=TARGET_SPACING, out_dtype=torch.bfloat16, device="cuda"):
    """The tensor part of reference `process_file` for model_type "ctclip" (:123-152).

    raw_hwd: the scan as stored, `[H, W, D]` (numpy array or tensor; int16 stays int16 on the wire, anything else goes as
    f32).  Returns `[1, D_t, H_t, W_t]` on the device in `out_dtype` (the reference returns the same tensor in f32)."""
    t = torch.as_tensor(raw_hwd)
    if t.dtype != torch.int16:
        t = t.to(torch.float32)
    t = t.to(device).contiguous()
    if not t.is_cuda:
        raise RuntimeError("process_volume: MI355X HIP path only (no CPU fallback)")
    H, W, D = (int(s) for s in t.shape)
    rD, rH, rW = resampled_shape((D, H, W), (z_spacing, xy_spacing, xy_spacing), target_spacing)
    oH, oW, oD = (int(s) for s in target_shape)
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("out_dtype must be bfloat16 or float32")
    out = torch.empty(1, oD, oH, oW, dtype=out_dtype, device=t.device)
    hip.ingest_volume(t, int(t.dtype == torch.int16), H, W, D, float(slope), float(intercept), rD, rH, rW, oD, oH, oW, -1.0,
                      out, int(out_dtype == torch.bfloat16))
    return out


def process_file(file_path, file_name, metadata_df, model_type, out_dtype=torch.bfloat16, device="cuda"):
    """reference :84-152 (same arguments; returns the `[1, D, H, W]` tensor on the device)."""
    if model_type != "ctclip":
        raise NotImplementedError("only the CT-CLIP ingest path is implemented (CT-Generate is a different model)")
    img = read_nii_data(file_path)
    if img is None:
        print(f"Read failure for {file_path}.")
        return None
    row = metadata_df[metadata_df["VolumeName"] == file_name]
    if row.empty:
        print(f"No metadata found for {file_name}.")
        return None
    try:
        slope = float(row["RescaleSlope"].iloc[0])
        intercept = float(row["RescaleIntercept"].iloc[0])
        xy_spacing = float(row["XYSpacing"].iloc[0][1:][:-2].split(",")[0])
        z_spacing = float(row["ZSpacing"].iloc[0])
    except Exception as e:
        print(f"Error processing metadata for {file_name}: {e}")
        return None
    return process_volume(img, slope, intercept, xy_spacing, z_spacing, out_dtype=out_dtype, device=device)
