"""TEST INFRASTRUCTURE -- CPU restatement of the homogeneous-texture path of
scripts/mpeg7Descriptors.py: gabor_filters (:287-305), radon_feature_extract (:308-365) and
mpeg7_homogeneus_texture (:277-284) with segment_generator (:28-38).

The third-party arithmetic is the real thing: scipy.ndimage.rotate and scipy.fftpack are
importable here (scipy 1.15; the reference pins no version and imports the same functions through
the old scipy.ndimage.interpolation alias), so this oracle is pinned by scipy itself for the
Radon / FFT part.  Python 2 -> 3: xrange -> range, nothing else changes.
"""
import numpy as np
import scipy.fftpack
import scipy.ndimage

CEN_FREQ = [3 / 4.0, 3 / 8.0, 3 / 16.0, 3 / 32.0, 3 / 64.0]          # const_data.py:1-4
OCT_BAND = [1 / 2.0, 1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0]
ANGULAR_BAND = [30, 30, 30, 30, 30, 30]
ANGULAR_CEN = [0, 30, 60, 90, 120, 150]
GABOR_RADIAL, GABOR_ANGULAR = 5, 6                                   # mpeg7Descriptors.py:13-14


def gabor_filters(num_rad, num_ang, plane_len):
    ang_w, rad_h = plane_len
    to_rad = lambda x: ((np.pi * x) / 180)
    angle_p = np.array([(2 * np.pi * ang) / ang_w for ang in range(ang_w)])
    radius_p = np.linspace(0, 1, rad_h)
    radius_p, angle_p = np.meshgrid(radius_p, angle_p)
    radius_p = radius_p.flatten()
    angle_p = angle_p.flatten()
    out = []
    for r_idx in range(num_rad):
        for a_idx in range(num_ang):
            rad_std = OCT_BAND[r_idx] / 2 * np.sqrt(2 * np.log(2))
            ang_std = to_rad(ANGULAR_BAND[a_idx]) / 2 * np.sqrt(2 * np.log(2))
            g = (np.exp((-((radius_p - 0.5) - CEN_FREQ[r_idx]) ** 2) / (2.0 * rad_std ** 2)) *
                 np.exp((-(angle_p - to_rad(ANGULAR_CEN[a_idx])) ** 2) / (2.0 * ang_std ** 2)))
            out.append(g.reshape(plane_len))
    return out


def radon_feature_extract(image, c_num=329):
    image = np.asarray(image, dtype=np.float32) / 255.0                          # :311
    height, width = image.shape
    if height > width:                                                            # :313-318
        image = np.concatenate((image, np.zeros((height, height - width))), axis=1)
    else:
        image = np.concatenate((image, np.zeros((width - height, width))), axis=0)
    height, width = image.shape
    fourier_res = width
    sinogram = np.array([np.sum(scipy.ndimage.rotate(image, np.rad2deg((np.pi * per) / c_num),
                                                     reshape=False, mode='constant', cval=0.0), axis=0)
                         for per in range(c_num)])                               # :323-331
    fft_rows = scipy.fftpack.fftshift(
        scipy.fftpack.fft(scipy.fftpack.ifftshift(sinogram, axes=1)), axes=1)    # :334-341
    angle_p = np.array([(2 * np.pi * ang) / c_num for ang in range(c_num)])
    radius_p = np.linspace(0, 1, fourier_res)
    radius_p, angle_p = np.meshgrid(radius_p, angle_p)
    feature_list = []
    with np.errstate(all="ignore"):
        for g_func in gabor_filters(GABOR_RADIAL, GABOR_ANGULAR, (c_num, fourier_res)):
            p = (g_func * radius_p * fft_rows) ** 2                              # complex square
            sum_p = np.sum(np.sum(p, axis=1))
            e = np.log10(1 + sum_p)
            q = ((g_func * radius_p * fft_rows) ** 2 - sum_p) ** 2
            sum_q = np.sqrt(np.sum(np.sum(q, axis=1)))
            d = np.log10(1 + sum_q)
            feature_list.append({'d': np.abs(d), 'e': np.abs(e)})
    feature_list.append(np.mean(image))
    feature_list.append(np.std(image))
    return feature_list


def homogeneous_texture(clusters, image_gray):
    """mpeg7_homogeneus_texture (:277-284): per segment the grey image with every OTHER segment
    zeroed (pixels outside every segment keep their grey level, :34-37)."""
    segments = set(clusters.flatten())
    segments.discard(-1)
    out = {}
    for s in segments:
        img = image_gray.copy()
        img[(clusters != s) & (clusters != -1)] = 0
        out[int(s)] = radon_feature_extract(img)
    return out
