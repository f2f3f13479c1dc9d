// Host-side multi-object tracker behind the C ABI (pp_tracker_*, pp_linear_sum_assignment).
//
// Association is strictly sequential in time and tiny (<= 100 x 100 per frame), so it stays on the
// host in float64 exactly like the reference (a GPU port would be launch-bound).
//
// mode 0 -- the tracker whose source is in the reference tree,
//   pose_pipeline/wrappers/deep_sort_yolov4/deep_sort/:
//     kalman_filter.py:14-217   8-state constant-velocity filter, std weights 1/20 and 1/160
//     linear_assignment.py:14-186  min_cost_matching (cost > max -> max + 1e-5, Hungarian),
//                                  matching_cascade by time_since_update, chi-square gating (9.4877)
//     iou_matching.py:7-84      IoU cost (1 - iou), INFTY for tracks with time_since_update > 1
//     nn_matching.py            cosine nearest-neighbour metric over each target's feature gallery
//     track.py / tracker.py:10-131  lifecycle (n_init 3, max_age 30, ids from 1)
//   pinned by tests/golden/deepsort.npz (traces produced by importing the reference).
// mode 1 -- mmtrack 0.x SortTracker as configured by
//   3rdparty/mmtracking/mot/deepsort/sort_faster-rcnn_fpn_4e_mot17-private-half.py (no ReID):
//   obj_score_thr 0.5, tracks seen in the previous frame vs detections, cost 1 - IoU (float32,
//   eps 1e-6), Hungarian, accept cost < 1 - match_iou_thr, ids from a running counter starting at 0.
//   mmtrack is not vendored: parity unpinned.
// The Hungarian solver restates scipy.optimize.linear_sum_assignment (rectangular LSAP, Crouse's
// shortest augmenting path, scipy/optimize/rectangular_lsap), including its tie-breaking, because
// bit-exact track ids depend on it; pinned by tests/golden/hungarian.npz.
//
// Third-party attribution: `solve_lsap` / `augmenting_path` below follow the structure and the identifier
// names (row4col, col4row, SR, SC, remaining, shortestPathCosts) of SciPy's
// scipy/optimize/rectangular_lsap/rectangular_lsap.cpp, "Copyright (c) 2019, PM Larsen (SciPy
// Developers)", distributed under the 3-clause BSD licence (redistribution in source and binary form
// permitted provided the copyright notice, the list of conditions and the disclaimer are retained; the
// SciPy developers' names may not be used to endorse derived products; provided "as is" without
// warranty).  It is not code of the reference repository (which only CALLS scipy, linear_assignment.py:55).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include "posepipe_hip.h"

void pp_set_error(const char* fmt, ...);

namespace {

// ---- scipy rectangular LSAP -------------------------------------------------------------------
typedef std::ptrdiff_t idx_t;

idx_t augmenting_path(idx_t nc, const double* cost, std::vector<double>& u, std::vector<double>& v,
                      std::vector<idx_t>& path, std::vector<idx_t>& row4col, std::vector<double>& spc, idx_t i,
                      std::vector<char>& SR, std::vector<char>& SC, std::vector<idx_t>& remaining, double* p_min) {
    double minVal = 0;
    idx_t num_remaining = nc;
    for (idx_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;   // reverse order: constant matrix -> identity
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INFINITY);
    idx_t sink = -1;
    while (sink == -1) {
        idx_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (idx_t it = 0; it < num_remaining; it++) {
            const idx_t j = remaining[it];
            const double r = minVal + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) {
                path[j] = i;
                spc[j] = r;
            }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        minVal = lowest;
        if (minVal == INFINITY) return -1;
        const idx_t j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = minVal;
    return sink;
}

// returns number of pairs (min(nr, nc)) or -1 when infeasible; pairs sorted by row
int lsap(const double* cost_in, idx_t nr, idx_t nc, std::vector<int>& rows, std::vector<int>& cols) {
    rows.clear();
    cols.clear();
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;
    std::vector<double> temp;
    const double* cost = cost_in;
    if (transpose) {
        temp.resize(nr * nc);
        for (idx_t i = 0; i < nr; i++)
            for (idx_t j = 0; j < nc; j++) temp[j * nr + i] = cost_in[i * nc + j];
        std::swap(nr, nc);
        cost = temp.data();
    }
    for (idx_t i = 0; i < nr * nc; i++)
        if (cost[i] != cost[i] || cost[i] == -INFINITY) return -1;
    std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
    std::vector<idx_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (idx_t cur = 0; cur < nr; cur++) {
        double minVal;
        const idx_t sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &minVal);
        if (sink < 0) return -1;
        u[cur] += minVal;
        for (idx_t i = 0; i < nr; i++)
            if (SR[i] && i != cur) u[i] += minVal - spc[col4row[i]];
        for (idx_t j = 0; j < nc; j++)
            if (SC[j]) v[j] -= minVal - spc[j];
        idx_t j = sink;
        while (true) {
            const idx_t i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        std::vector<idx_t> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](idx_t a, idx_t b) { return col4row[a] < col4row[b]; });
        for (idx_t k : order) {
            rows.push_back((int)col4row[k]);
            cols.push_back((int)k);
        }
    } else {
        for (idx_t i = 0; i < nr; i++) {
            rows.push_back((int)i);
            cols.push_back((int)col4row[i]);
        }
    }
    return (int)rows.size();
}

// ---- Kalman filter (kalman_filter.py) -------------------------------------------------------------
struct KF {
    static constexpr double W_POS = 1.0 / 20, W_VEL = 1.0 / 160;

    static void initiate(const double z[4], double mean[8], double cov[64]) {
        for (int i = 0; i < 4; i++) {
            mean[i] = z[i];
            mean[4 + i] = 0;
        }
        const double std_[8] = {2 * W_POS * z[3], 2 * W_POS * z[3], 1e-2, 2 * W_POS * z[3],
                                10 * W_VEL * z[3], 10 * W_VEL * z[3], 1e-5, 10 * W_VEL * z[3]};
        std::fill(cov, cov + 64, 0.0);
        for (int i = 0; i < 8; i++) cov[i * 8 + i] = std_[i] * std_[i];
    }

    static void predict(double mean[8], double cov[64]) {
        const double std_[8] = {W_POS * mean[3], W_POS * mean[3], 1e-2, W_POS * mean[3],
                                W_VEL * mean[3], W_VEL * mean[3], 1e-5, W_VEL * mean[3]};
        // mean = F mean, F = I + shift(dt = 1)
        for (int i = 0; i < 4; i++) mean[i] += mean[4 + i];
        // cov = F cov F^T + Q
        double fp[64];
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) fp[i * 8 + j] = cov[i * 8 + j] + (i < 4 ? cov[(i + 4) * 8 + j] : 0.0);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) cov[i * 8 + j] = fp[i * 8 + j] + (j < 4 ? fp[i * 8 + j + 4] : 0.0);
        for (int i = 0; i < 8; i++) cov[i * 8 + i] += std_[i] * std_[i];
    }

    static void project(const double mean[8], const double cov[64], double pm[4], double pc[16]) {
        const double std_[4] = {W_POS * mean[3], W_POS * mean[3], 1e-1, W_POS * mean[3]};
        for (int i = 0; i < 4; i++) {
            pm[i] = mean[i];
            for (int j = 0; j < 4; j++) pc[i * 4 + j] = cov[i * 8 + j];
            pc[i * 4 + i] += std_[i] * std_[i];
        }
    }

    static bool cholesky4(const double a[16], double l[16]) {
        std::fill(l, l + 16, 0.0);
        for (int j = 0; j < 4; j++) {
            double s = a[j * 4 + j];
            for (int k = 0; k < j; k++) s -= l[j * 4 + k] * l[j * 4 + k];
            if (s <= 0) return false;
            l[j * 4 + j] = std::sqrt(s);
            for (int i = j + 1; i < 4; i++) {
                double t = a[i * 4 + j];
                for (int k = 0; k < j; k++) t -= l[i * 4 + k] * l[j * 4 + k];
                l[i * 4 + j] = t / l[j * 4 + j];
            }
        }
        return true;
    }

    static void update(double mean[8], double cov[64], const double z[4]) {
        double pm[4], pc[16], l[16];
        project(mean, cov, pm, pc);
        cholesky4(pc, l);
        // kalman_gain (8x4) = cov[:, :4] * pc^-1   (solve pc * K^T = (cov H^T)^T)
        double K[32];
        for (int r = 0; r < 8; r++) {
            double y[4], x[4];
            for (int i = 0; i < 4; i++) {            // forward: l y = b, b = cov[r, :4]
                double s = cov[r * 8 + i];
                for (int k = 0; k < i; k++) s -= l[i * 4 + k] * y[k];
                y[i] = s / l[i * 4 + i];
            }
            for (int i = 3; i >= 0; i--) {           // backward: l^T x = y
                double s = y[i];
                for (int k = i + 1; k < 4; k++) s -= l[k * 4 + i] * x[k];
                x[i] = s / l[i * 4 + i];
            }
            for (int i = 0; i < 4; i++) K[r * 4 + i] = x[i];
        }
        double innov[4];
        for (int i = 0; i < 4; i++) innov[i] = z[i] - pm[i];
        for (int r = 0; r < 8; r++) {
            double s = 0;
            for (int i = 0; i < 4; i++) s += innov[i] * K[r * 4 + i];
            mean[r] += s;
        }
        // cov -= K pc K^T
        double kp[32];
        for (int r = 0; r < 8; r++)
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int i = 0; i < 4; i++) s += K[r * 4 + i] * pc[i * 4 + j];
                kp[r * 4 + j] = s;
            }
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) {
                double s = 0;
                for (int i = 0; i < 4; i++) s += kp[r * 4 + i] * K[c * 4 + i];
                cov[r * 8 + c] -= s;
            }
    }

    // squared Mahalanobis distance of measurement z to the projected state
    static double gating_distance(const double mean[8], const double cov[64], const double z[4]) {
        double pm[4], pc[16], l[16];
        project(mean, cov, pm, pc);
        cholesky4(pc, l);
        double y[4], acc = 0;
        for (int i = 0; i < 4; i++) {
            double s = z[i] - pm[i];
            for (int k = 0; k < i; k++) s -= l[i * 4 + k] * y[k];
            y[i] = s / l[i * 4 + i];
            acc += y[i] * y[i];
        }
        return acc;
    }
};

enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };
constexpr double INFTY_COST = 1e5;
constexpr double CHI2INV95_4 = 9.4877;

struct Track {
    double mean[8], cov[64];
    int64_t id;
    int hits = 1, age = 1, time_since_update = 0, state = TENTATIVE;
    std::vector<std::vector<double>> features;   // pending features (moved to the gallery on update)
    void to_tlwh(double out[4]) const {
        out[2] = mean[2] * mean[3];
        out[3] = mean[3];
        out[0] = mean[0] - out[2] / 2;
        out[1] = mean[1] - out[3] / 2;
    }
};

void tlwh_to_xyah(const double t[4], double z[4]) {
    z[0] = t[0] + t[2] / 2;
    z[1] = t[1] + t[3] / 2;
    z[2] = t[2] / t[3];
    z[3] = t[3];
}

double iou_tlwh(const double a[4], const double b[4]) {
    const double tlx = std::max(a[0], b[0]), tly = std::max(a[1], b[1]);
    const double brx = std::min(a[0] + a[2], b[0] + b[2]), bry = std::min(a[1] + a[3], b[1] + b[3]);
    const double w = std::max(0.0, brx - tlx), h = std::max(0.0, bry - tly);
    const double inter = w * h;
    return inter / (a[2] * a[3] + b[2] * b[3] - inter);
}

}  // namespace

struct pp_tracker {
    int mode = 0;
    double max_iou_distance = 0.7, max_cosine_distance = 0.3;
    int max_age = 30, n_init = 3, feat_dim = 0;
    int64_t next_id = 1;
    std::vector<Track> tracks;
    // nn metric gallery: target id -> list of unit feature vectors (budget None)
    std::vector<std::pair<int64_t, std::vector<std::vector<double>>>> gallery;
    // mode 1 state
    double obj_score_thr = 0.5, match_iou_thr = 0.5;
    struct SortTrack { int64_t id; float box[4]; int last_frame; };
    std::vector<SortTrack> sort_tracks;
    int frame_id = 0;

    std::vector<std::vector<double>>* find_gallery(int64_t id) {
        for (auto& g : gallery)
            if (g.first == id) return &g.second;
        return nullptr;
    }
};

namespace {

struct Det {
    double tlwh[4];
    double conf;
    const double* feat;
};

typedef std::vector<std::pair<int, int>> Matches;

// linear_assignment.min_cost_matching on a prepared cost matrix
void min_cost_matching(std::vector<double>& cost, double max_distance, const std::vector<int>& trk_idx,
                       const std::vector<int>& det_idx, Matches& matches, std::vector<int>& um_trk,
                       std::vector<int>& um_det) {
    matches.clear();
    um_trk.clear();
    um_det.clear();
    const int nt = (int)trk_idx.size(), nd = (int)det_idx.size();
    if (nt == 0 || nd == 0) {
        um_trk = trk_idx;
        um_det = det_idx;
        return;
    }
    for (double& c : cost)
        if (c > max_distance) c = max_distance + 1e-5;
    std::vector<int> rows, cols;
    lsap(cost.data(), nt, nd, rows, cols);
    std::vector<char> col_used(nd, 0), row_used(nt, 0);
    for (size_t k = 0; k < rows.size(); k++) {
        row_used[rows[k]] = 1;
        col_used[cols[k]] = 1;
    }
    for (int c = 0; c < nd; c++)
        if (!col_used[c]) um_det.push_back(det_idx[c]);
    for (int r = 0; r < nt; r++)
        if (!row_used[r]) um_trk.push_back(trk_idx[r]);
    for (size_t k = 0; k < rows.size(); k++) {
        const int r = rows[k], c = cols[k];
        if (cost[(size_t)r * nd + c] > max_distance) {
            um_trk.push_back(trk_idx[r]);
            um_det.push_back(det_idx[c]);
        } else {
            matches.emplace_back(trk_idx[r], det_idx[c]);
        }
    }
}

// nn_matching._nn_cosine_distance against a gallery, then gate_cost_matrix
void gated_metric(pp_tracker* t, const std::vector<Det>& dets, const std::vector<int>& trk_idx,
                  const std::vector<int>& det_idx, std::vector<double>& cost) {
    const int nt = (int)trk_idx.size(), nd = (int)det_idx.size(), D = t->feat_dim;
    cost.assign((size_t)nt * nd, 0.0);
    // detection features normalised once (a / ||a||)
    std::vector<double> dn((size_t)nd * D);
    for (int c = 0; c < nd; c++) {
        const double* f = dets[det_idx[c]].feat;
        double s = 0;
        for (int k = 0; k < D; k++) s += f[k] * f[k];
        s = std::sqrt(s);
        for (int k = 0; k < D; k++) dn[(size_t)c * D + k] = f[k] / s;
    }
    for (int r = 0; r < nt; r++) {
        const Track& trk = t->tracks[trk_idx[r]];
        auto* gal = t->find_gallery(trk.id);
        for (int c = 0; c < nd; c++) {
            double best = INFINITY;
            if (gal) {
                for (const auto& g : *gal) {
                    double gs = 0;
                    for (int k = 0; k < D; k++) gs += g[k] * g[k];
                    gs = std::sqrt(gs);
                    double dot = 0;
                    for (int k = 0; k < D; k++) dot += (g[k] / gs) * dn[(size_t)c * D + k];
                    best = std::min(best, 1.0 - dot);
                }
            }
            cost[(size_t)r * nd + c] = best;
        }
        for (int c = 0; c < nd; c++) {
            double z[4];
            tlwh_to_xyah(dets[det_idx[c]].tlwh, z);
            if (KF::gating_distance(trk.mean, trk.cov, z) > CHI2INV95_4) cost[(size_t)r * nd + c] = INFTY_COST;
        }
    }
}

void iou_cost(pp_tracker* t, const std::vector<Det>& dets, const std::vector<int>& trk_idx,
              const std::vector<int>& det_idx, std::vector<double>& cost) {
    const int nt = (int)trk_idx.size(), nd = (int)det_idx.size();
    cost.assign((size_t)nt * nd, 0.0);
    for (int r = 0; r < nt; r++) {
        const Track& trk = t->tracks[trk_idx[r]];
        if (trk.time_since_update > 1) {
            for (int c = 0; c < nd; c++) cost[(size_t)r * nd + c] = INFTY_COST;
            continue;
        }
        double tb[4];
        trk.to_tlwh(tb);
        for (int c = 0; c < nd; c++) cost[(size_t)r * nd + c] = 1.0 - iou_tlwh(tb, dets[det_idx[c]].tlwh);
    }
}

int deepsort_step(pp_tracker* t, const std::vector<Det>& dets) {
    // Tracker.predict
    for (auto& trk : t->tracks) {
        KF::predict(trk.mean, trk.cov);
        trk.age += 1;
        trk.time_since_update += 1;
    }
    // Tracker._match
    std::vector<int> confirmed, unconfirmed;
    for (int i = 0; i < (int)t->tracks.size(); i++) (t->tracks[i].state == CONFIRMED ? confirmed : unconfirmed).push_back(i);
    // matching_cascade over confirmed tracks
    Matches matches_a;
    std::vector<int> um_det((int)dets.size());
    std::iota(um_det.begin(), um_det.end(), 0);
    std::vector<double> cost;
    for (int level = 0; level < t->max_age; level++) {
        if (um_det.empty()) break;
        std::vector<int> lvl;
        for (int k : confirmed)
            if (t->tracks[k].time_since_update == 1 + level) lvl.push_back(k);
        if (lvl.empty()) continue;
        gated_metric(t, dets, lvl, um_det, cost);
        Matches m;
        std::vector<int> ut, ud;
        min_cost_matching(cost, t->max_cosine_distance, lvl, um_det, m, ut, ud);
        um_det = ud;
        matches_a.insert(matches_a.end(), m.begin(), m.end());
    }
    // unmatched_tracks = list(set(track_indices) - set(matched)): CPython small-int sets iterate ascending
    std::vector<int> um_trk_a;
    {
        std::set<int> s(confirmed.begin(), confirmed.end());
        for (auto& m : matches_a) s.erase(m.first);
        um_trk_a.assign(s.begin(), s.end());
    }
    std::vector<int> iou_cand = unconfirmed;
    std::vector<int> um_trk_a2;
    for (int k : um_trk_a) (t->tracks[k].time_since_update == 1 ? iou_cand : um_trk_a2).push_back(k);
    iou_cost(t, dets, iou_cand, um_det, cost);
    Matches matches_b;
    std::vector<int> um_trk_b, um_det_b;
    min_cost_matching(cost, t->max_iou_distance, iou_cand, um_det, matches_b, um_trk_b, um_det_b);
    Matches matches = matches_a;
    matches.insert(matches.end(), matches_b.begin(), matches_b.end());
    std::set<int> um_trk(um_trk_a2.begin(), um_trk_a2.end());
    um_trk.insert(um_trk_b.begin(), um_trk_b.end());
    // Tracker.update
    for (auto& m : matches) {
        Track& trk = t->tracks[m.first];
        double z[4];
        tlwh_to_xyah(dets[m.second].tlwh, z);
        KF::update(trk.mean, trk.cov, z);
        trk.features.emplace_back(dets[m.second].feat, dets[m.second].feat + t->feat_dim);
        trk.hits += 1;
        trk.time_since_update = 0;
        if (trk.state == TENTATIVE && trk.hits >= t->n_init) trk.state = CONFIRMED;
    }
    for (int k : um_trk) {
        Track& trk = t->tracks[k];
        if (trk.state == TENTATIVE) trk.state = DELETED;
        else if (trk.time_since_update > t->max_age) trk.state = DELETED;
    }
    for (int d : um_det_b) {
        Track trk;
        double z[4];
        tlwh_to_xyah(dets[d].tlwh, z);
        KF::initiate(z, trk.mean, trk.cov);
        trk.id = t->next_id++;
        trk.features.emplace_back(dets[d].feat, dets[d].feat + t->feat_dim);
        t->tracks.push_back(std::move(trk));
    }
    t->tracks.erase(std::remove_if(t->tracks.begin(), t->tracks.end(), [](const Track& k) { return k.state == DELETED; }),
                    t->tracks.end());
    // metric.partial_fit: append pending features of confirmed tracks, keep only active targets
    std::vector<std::pair<int64_t, std::vector<std::vector<double>>>> ng;
    for (auto& trk : t->tracks) {
        if (trk.state != CONFIRMED) continue;
        auto* g = t->find_gallery(trk.id);
        std::vector<std::vector<double>> feats = g ? *g : std::vector<std::vector<double>>();
        for (auto& f : trk.features) feats.push_back(f);
        trk.features.clear();
        ng.emplace_back(trk.id, std::move(feats));
    }
    t->gallery.swap(ng);
    return PP_OK;
}

}  // namespace

extern "C" {

int pp_linear_sum_assignment(const double* cost, int n_rows, int n_cols, int32_t* rows, int32_t* cols,
                             int32_t* n_pairs) {
    if (!cost || !rows || !cols || !n_pairs || n_rows < 0 || n_cols < 0) {
        pp_set_error("pp_linear_sum_assignment: bad argument");
        return PP_ERR_ARG;
    }
    std::vector<int> r, c;
    const int n = lsap(cost, n_rows, n_cols, r, c);
    if (n < 0) {
        pp_set_error("pp_linear_sum_assignment: cost matrix is infeasible (NaN / -inf / all-inf row)");
        return PP_ERR_ARG;
    }
    for (int i = 0; i < n; i++) {
        rows[i] = r[i];
        cols[i] = c[i];
    }
    *n_pairs = n;
    return PP_OK;
}

int pp_tracker_create(int mode, int feat_dim, double max_iou_distance, double max_cosine_distance, int max_age,
                      int n_init, pp_tracker** out) {
    if (!out || (mode != 0 && mode != 1) || feat_dim < 0 || (mode == 0 && feat_dim < 1)) {
        pp_set_error("pp_tracker_create: bad argument (mode 0 needs feat_dim >= 1)");
        return PP_ERR_ARG;
    }
    pp_tracker* t = new pp_tracker();
    t->mode = mode;
    t->feat_dim = feat_dim;
    if (mode == 0) {
        t->max_iou_distance = max_iou_distance;
        t->max_cosine_distance = max_cosine_distance;
        t->max_age = max_age;
        t->n_init = n_init;
        t->next_id = 1;
    } else {
        t->match_iou_thr = max_iou_distance;     // mode 1: accept cost < 1 - match_iou_thr
        t->obj_score_thr = max_cosine_distance;  // mode 1: detection score threshold
        t->next_id = 0;
    }
    *out = t;
    return PP_OK;
}

void pp_tracker_destroy(pp_tracker* t) { delete t; }

int pp_tracker_step(pp_tracker* t, const double* dets_tlwh, const double* conf, const double* feats, int n_det,
                    int cap, int64_t* track_id, double* tlwh, int32_t* info, int32_t* n_out) {
    if (!t || n_det < 0 || (n_det > 0 && (!dets_tlwh || !conf)) || !n_out) {
        pp_set_error("pp_tracker_step: bad argument");
        return PP_ERR_ARG;
    }
    *n_out = 0;
    if (t->mode == 0) {
        if (n_det > 0 && !feats) {
            pp_set_error("pp_tracker_step: mode 0 needs appearance features");
            return PP_ERR_ARG;
        }
        std::vector<Det> dets(n_det);
        for (int i = 0; i < n_det; i++) {
            memcpy(dets[i].tlwh, dets_tlwh + 4 * i, 4 * sizeof(double));
            dets[i].conf = conf[i];
            dets[i].feat = feats + (size_t)i * t->feat_dim;
        }
        deepsort_step(t, dets);
        // every live track, as wrappers/deep_sort_yolov4/parser.py:76-86 emits them
        int n = 0;
        for (const auto& trk : t->tracks) {
            if (n >= cap) break;
            if (track_id) track_id[n] = trk.id;
            if (tlwh) trk.to_tlwh(tlwh + 4 * n);
            if (info) {
                info[4 * n + 0] = trk.state;
                info[4 * n + 1] = trk.hits;
                info[4 * n + 2] = trk.age;
                info[4 * n + 3] = trk.time_since_update;
            }
            n++;
        }
        *n_out = n;
        return PP_OK;
    }
    // ---- mode 1: mmtrack SortTracker without ReID ----------------------------------------------
    std::vector<int> keep;
    for (int i = 0; i < n_det; i++)
        if ((float)conf[i] > (float)t->obj_score_thr) keep.push_back(i);
    const int nk = (int)keep.size();
    std::vector<int64_t> ids(nk, -1);
    std::vector<float> boxes((size_t)nk * 4);   // x1 y1 x2 y2 float32 (mmdet tensors)
    for (int k = 0; k < nk; k++) {
        const double* b = dets_tlwh + 4 * keep[k];
        boxes[4 * k + 0] = (float)b[0];
        boxes[4 * k + 1] = (float)b[1];
        boxes[4 * k + 2] = (float)b[2];   // mode 1 takes the detector's x1 y1 x2 y2 rows as they are
        boxes[4 * k + 3] = (float)b[3];
    }
    if (!t->sort_tracks.empty() && nk > 0) {
        std::vector<int> active;
        for (int i = 0; i < (int)t->sort_tracks.size(); i++)
            if (t->sort_tracks[i].last_frame == t->frame_id - 1) active.push_back(i);
        if (!active.empty()) {
            std::vector<double> dist((size_t)active.size() * nk);
            for (size_t r = 0; r < active.size(); r++) {
                const float* a = t->sort_tracks[active[r]].box;
                for (int c = 0; c < nk; c++) {
                    const float* b = &boxes[4 * c];
                    const float area1 = (a[2] - a[0]) * (a[3] - a[1]);
                    const float area2 = (b[2] - b[0]) * (b[3] - b[1]);
                    const float w = std::max(std::min(a[2], b[2]) - std::max(a[0], b[0]), 0.f);
                    const float h = std::max(std::min(a[3], b[3]) - std::max(a[1], b[1]), 0.f);
                    const float overlap = w * h;
                    const float uni = std::max(area1 + area2 - overlap, 1e-6f);
                    dist[r * nk + c] = (double)(1.0f - overlap / uni);
                }
            }
            std::vector<int> rows, cols;
            lsap(dist.data(), (idx_t)active.size(), nk, rows, cols);
            for (size_t k = 0; k < rows.size(); k++)
                if (dist[(size_t)rows[k] * nk + cols[k]] < 1.0 - t->match_iou_thr)
                    ids[cols[k]] = t->sort_tracks[active[rows[k]]].id;
        }
    }
    for (int k = 0; k < nk; k++)
        if (ids[k] < 0) ids[k] = t->next_id++;
    // update memo: matched tracks refresh, new ones are appended; tracks unseen this frame can never be
    // re-associated without ReID (only last-frame tracks are candidates), so they are dropped
    std::vector<pp_tracker::SortTrack> next;
    for (int k = 0; k < nk; k++) {
        pp_tracker::SortTrack s;
        s.id = ids[k];
        memcpy(s.box, &boxes[4 * k], sizeof(s.box));
        s.last_frame = t->frame_id;
        next.push_back(s);
    }
    t->sort_tracks.swap(next);
    t->frame_id++;
    int n = 0;
    for (int k = 0; k < nk && n < cap; k++, n++) {
        if (track_id) track_id[n] = ids[k];
        if (tlwh) memcpy(tlwh + 4 * n, dets_tlwh + 4 * keep[k], 4 * sizeof(double));
        if (info) {
            info[4 * n + 0] = CONFIRMED;
            info[4 * n + 1] = keep[k];     // index of the source detection
            info[4 * n + 2] = 0;
            info[4 * n + 3] = 0;
        }
    }
    *n_out = n;
    return PP_OK;
}

int pp_tracker_dump(pp_tracker* t, int cap, int64_t* ids, int32_t* state4, double* mean8, double* cov64,
                    int32_t* n_out) {
    if (!t || !n_out) {
        pp_set_error("pp_tracker_dump: bad argument");
        return PP_ERR_ARG;
    }
    int n = 0;
    for (const auto& trk : t->tracks) {
        if (n >= cap) break;
        if (ids) ids[n] = trk.id;
        if (state4) {
            state4[4 * n] = trk.state;
            state4[4 * n + 1] = trk.hits;
            state4[4 * n + 2] = trk.age;
            state4[4 * n + 3] = trk.time_since_update;
        }
        if (mean8) memcpy(mean8 + 8 * n, trk.mean, sizeof(trk.mean));
        if (cov64) memcpy(cov64 + 64 * n, trk.cov, sizeof(trk.cov));
        n++;
    }
    *n_out = n;
    return PP_OK;
}

// standalone Kalman entry points (parity tests against kalman_filter.py)
int pp_kalman_initiate(const double* z4, double* mean8, double* cov64) {
    KF::initiate(z4, mean8, cov64);
    return PP_OK;
}
int pp_kalman_predict(double* mean8, double* cov64) {
    KF::predict(mean8, cov64);
    return PP_OK;
}
int pp_kalman_update(double* mean8, double* cov64, const double* z4) {
    KF::update(mean8, cov64, z4);
    return PP_OK;
}
int pp_kalman_gating_distance(const double* mean8, const double* cov64, const double* z4, int n, double* out) {
    for (int i = 0; i < n; i++) out[i] = KF::gating_distance(mean8, cov64, z4 + 4 * i);
    return PP_OK;
}

}  // extern "C"
