"""Drop-in for scripts/metric_methods.py (host-side scalar helpers, same results)."""
import copy

import numpy as np


def euclidean_dist(col_one, col_two):
    """metric_methods.py:5-6."""
    return np.sqrt((col_one[0] - col_two[0]) ** 2 + (col_one[1] - col_two[1]) ** 2 +
                   (col_one[2] - col_two[2]) ** 2)


def meas_sim_texture_desc(obs_one, obs_two):
    """metric_methods.py:9-23."""
    dist = 0
    weight = 20
    for feat_o, feat_t in zip(obs_one, obs_two):
        if type(feat_o) != type(feat_t):
            raise ValueError("Observation features should have same type!")
        if isinstance(feat_o, dict):
            dist += abs(feat_o['e'] - feat_t['e']) ** 2 * weight / 2.0
            dist += abs(feat_o['d'] - feat_t['d']) ** 2 * weight
        elif isinstance(feat_o, float):
            dist += (feat_o - feat_t) ** 2 * weight
        else:
            raise ValueError("Wrong attribute type!")
    return np.sqrt(dist)


def meas_sim_shape(obs_one, obs_two):
    """metric_methods.py:26-32."""
    dist = 0
    for n_feat_o, n_feat_t in zip(obs_one, obs_two):
        for s_om_o, s_om_t in zip(n_feat_o, n_feat_t):
            dist += (s_om_o - s_om_t) ** 2
    return np.sqrt(dist)


def meas_sim_col(obs_one, obs_two):
    """metric_methods.py:35-55."""
    percentage_weight = 600
    spatial_coh_weight = 600
    obo = copy.deepcopy(list(obs_one.values()))
    obt = copy.deepcopy(list(obs_two.values()))
    dist = 0
    for cluster in obo:
        sim_lst = [euclidean_dist(cluster[0], cl_sus[0]) for cl_sus in obt]
        min_val = min(sim_lst)
        min_idx = sim_lst.index(min_val)
        dist += min_val
        dist += euclidean_dist(cluster[1], obt[min_idx][1])
        dist += (cluster[2] - obt[min_idx][2]) ** 2 * percentage_weight
        dist += (cluster[3] - obt[min_idx][3]) ** 2 * spatial_coh_weight
        obt = [elem for idx, elem in enumerate(obt) if idx != min_idx]
    return np.sqrt(dist)


def spatial_distance(rel_o, rel_t):
    """metric_methods.py:58-59."""
    return np.sqrt(np.sum(np.square(rel_t - rel_o))) / np.sqrt(8)
