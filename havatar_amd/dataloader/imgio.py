"""Image file I/O for the dataset readers and the harnesses.  The reference goes through OpenCV (`cv2.imread` +
`COLOR_BGR2RGB`, `cv2.resize`, `cv2.imwrite`); this image has no cv2, so the few calls the path needs are restated on PIL /
numpy with OpenCV's conventions (uint8 RGB in memory, `INTER_AREA` = box mean for integer shrink factors, `INTER_LINEAR` =
half-pixel-centre bilinear with round-half-up to uint8)."""
import numpy as np


def _pil():
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("havatar_amd.dataloader needs Pillow to read/write PNG files") from e
    return Image


def imread_rgb(path):
    """uint8 [H,W,3] RGB == cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB) (cv2.imread drops alpha, expands grey)."""
    with _pil().open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


def imwrite_rgb(path, img):
    """Write uint8 [H,W,3] RGB == cv2.imwrite(path, cv2.cvtColor(img, cv2.COLOR_RGB2BGR))."""
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("imwrite_rgb expects uint8 [H,W,3]")
    _pil().fromarray(img, "RGB").save(path, compress_level=3)        # deflate level of cv2.imwrite's default; pixels are lossless either way


def resize_area(img, fx):
    """cv2.resize(img, (0,0), fx=fx, fy=fx, interpolation=cv2.INTER_AREA) for fx = 1/k, k integer: k x k box mean,
    rounded half up for uint8 (OpenCV's saturate_cast), exact for float arrays."""
    k = int(round(1.0 / fx))
    if k < 1 or abs(1.0 / k - fx) > 1e-9:
        raise NotImplementedError("area down-sampling is implemented for 1/integer factors (the reference uses 0.25); got %r" % fx)
    if k == 1:
        return img
    H, W = img.shape[:2]
    if H % k or W % k:
        raise NotImplementedError("area down-sampling needs sizes divisible by %d, got %dx%d" % (k, H, W))
    x = img.reshape(H // k, k, W // k, k, *img.shape[2:]).astype(np.float64).mean(axis=(1, 3))
    if img.dtype == np.uint8:
        return np.clip(np.floor(x + 0.5), 0, 255).astype(np.uint8)
    return x.astype(img.dtype)


def resize_linear(img, res):
    """cv2.resize(img, (res,res), interpolation=cv2.INTER_LINEAR): bilinear, half-pixel centres, clamped borders."""
    H, W = img.shape[:2]
    if (H, W) == (res, res):
        return img

    def taps(n_in, n_out):
        c = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(c).astype(np.int64)
        f = c - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f

    y0, y1, fy = taps(H, res)
    x0, x1, fx = taps(W, res)
    a = img.astype(np.float64)
    top = a[y0][:, x0] * (1 - fx)[None, :, None] + a[y0][:, x1] * fx[None, :, None]
    bot = a[y1][:, x0] * (1 - fx)[None, :, None] + a[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    if img.dtype == np.uint8:
        return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)


def erode_rect(mask_u8, k):
    """cv2.erode(mask, cv2.getStructuringElement(cv2.MORPH_RECT, (k,k))): k x k minimum filter, border treated as +inf."""
    from scipy import ndimage
    return ndimage.minimum_filter(mask_u8, size=(k, k), mode="constant", cval=255)
