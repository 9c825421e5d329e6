"""TEST INFRASTRUCTURE -- CPU restatement of the region-shape path of
scripts/mpeg7Descriptors.py: cart2radial (:144-186), art_transform (:233-257) and
mpeg7_region_shape (:259-275), with segment_generator (:28-38).

cart2radial is the literal per-pixel loop (Python 2 semantics restated: `/` between ints is
floor division), so it is only for small images.  Parity unpinned by the reference (no
fixtures; needs OpenCV for the grey image): the grey image is an input here.
"""
import numpy as np


def cart2radial(image):
    """One-channel, interp=True branch of mpeg7Descriptors.py:144-186."""
    height, width = image.shape
    blank_image = np.zeros((height, width), np.uint8)
    max_rad = np.sqrt(width ** 2 + height ** 2) / 2.0                    # :153
    r_scale = max_rad / width
    ang_scale = (2 * np.pi) / height
    for y_a in range(height):
        for x_a in range(width):
            angle = y_a * ang_scale
            radius = x_a * r_scale
            polar_x = radius * np.cos(angle) + width / 2.0               # :161
            polar_y = radius * np.sin(angle) + height / 2.0
            x_r = [int(polar_x), int(polar_x) + 2]                       # :165-171
            y_r = [int(polar_y), int(polar_y) + 2]
            x_r[0] = x_r[0] if x_r[0] >= 0 else 0
            x_r[1] = x_r[1] if x_r[1] <= width else width
            y_r[0] = y_r[0] if y_r[0] >= 0 else 0
            y_r[1] = y_r[1] if y_r[1] <= height else height
            window = image[y_r[0]:y_r[1], x_r[0]:x_r[1]]                 # Python slice rules
            values = [int(v) for row in window for v in row.tolist()]    # :173-176
            if len(values) == 0:
                continue
            blank_image[y_a, x_a] = sum(values) // len(values)           # :179, Py2 int division
    return blank_image


def art_transform(image, polar_image=None):
    """mpeg7Descriptors.py:233-257 (numpy, as written)."""
    coeffs_n, coeffs_m = (3, 12)
    height, width = image.shape
    ang_scale = (2 * np.pi) / float(height)
    r_scale = 1 / float(width)
    if polar_image is None:
        polar_image = cart2radial(image)
    result_art = np.zeros((coeffs_n, coeffs_m), dtype=np.float64)
    angle_mat = ang_scale * (np.ones((width, 1)) * np.arange(height)).transpose()
    radius_mat = r_scale * np.arange(width)
    for n in range(coeffs_n):
        for m in range(coeffs_m):
            a_m = (1 / (2 * np.pi)) * np.exp(1j * m * angle_mat)
            if n == 0:
                r_n = np.ones((1, width))
            else:
                r_n = 2 * np.cos(np.pi * n * radius_mat)
            r_n = r_n * np.eye(width)
            v_base = np.dot(a_m, r_n)
            image_mult = polar_image * v_base * r_scale * (np.ones((height, 1)) * np.arange(width))
            high_sum = np.sum(image_mult, axis=1, dtype=np.complex128) * r_scale
            art_coeff = np.sum(high_sum, dtype=np.complex128) * ang_scale
            result_art[n, m] = np.abs(art_coeff)
    if result_art[0, 0] != 0:
        result_art /= result_art[0, 0]
    return result_art.tolist()


def segment_value_image(clusters, image_gray, segment_idx):
    """segment_generator (:28-38) + BGR2GRAY + `new_image[idx] = 1.0` (:264-266): other segments
    0, the segment 1, pixels outside every segment (-1) keep their grey level."""
    out = image_gray.copy()
    out[(clusters != segment_idx) & (clusters != -1)] = 0
    out[clusters == segment_idx] = 1
    return out


def region_shape(clusters, image_gray):
    """mpeg7_region_shape (:259-275): the transform runs on the whole frame (the crop computed
    at :270-272 is never used)."""
    segments = set(clusters.flatten())
    segments.discard(-1)
    return {int(s): art_transform(segment_value_image(clusters, image_gray, s)) for s in segments}
