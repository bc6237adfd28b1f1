"""CPU restatement of the reference's StereoRefine state machine (poselib/source/stereo_pose_refinement.cpp), test infrastructure only.

Every numerical primitive comes from the CPU oracle (oracle/pose_oracle.c through tests/oracle_lib.py); this file restates the control flow
around them for the estimators of the hot path (RobMethod RANSAC / LMEDS, checkPoolPoseRobust = 1, no refinement, no BA), the same scope
include/matchinglib_poselib/stereo_pose_refinement.h documents.  Line numbers cite the reference file.

The pool is a plain list of dicts in insertion order (the reference keeps a std::list plus an index map plus row numbers into two
matrices; the position in the list plays all three roles here).
"""
import math

import numpy as np

F32 = np.float32


class Cfg:
    def __init__(self, **kw):
        self.th_pix_user = 0.8
        self.RobMethod = "RANSAC"
        self.refineRTold = False
        self.minStartAggInlRat = 0.2
        self.relInlRatThLast = 0.35
        self.relInlRatThNew = 0.2
        self.minInlierRatSkip = 0.38
        self.relMinInlierRatSkip = 0.7
        self.maxSkipPairs = 5
        self.minInlierRatioReInit = 0.6
        self.minPtsDistance = 3.0
        self.maxPoolCorrespondences = 30000
        self.minContStablePoses = 3
        self.absThRankingStable = 0.075
        self.useRANSAC_fewMatches = False
        self.minNormDistStable = 0.5
        self.raiseSkipCnt = 0
        self.maxRat3DPtsFar = 0.5
        self.maxDist3DPtsZ = 50.0
        self.autoTH = False
        self.checkPoolPoseRobust = 1       # the struct's default is 3; 1 = always re-estimate robustly (the tests' historical setting)
        self.refineRTold_CorrPool = False
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)

    def as_doubles(self):
        return np.array([self.th_pix_user, self.minStartAggInlRat, self.relInlRatThLast, self.relInlRatThNew, self.minInlierRatSkip,
                         self.relMinInlierRatSkip, self.maxSkipPairs, self.minInlierRatioReInit, self.minPtsDistance,
                         self.maxPoolCorrespondences, self.minContStablePoses, self.absThRankingStable, float(self.useRANSAC_fewMatches),
                         self.minNormDistStable, self.raiseSkipCnt, self.maxRat3DPtsFar, self.maxDist3DPtsZ, float(self.refineRTold),
                         float(self.checkPoolPoseRobust), float(self.refineRTold_CorrPool), float(self.autoTH)], np.float64)


def full_stats(vals, rej_quartiles, round_std=True):
    """getStatsfromVec (pose_helper.cpp:358-413) -> (medErr, arithErr, arithStd, medStd)."""
    v = np.sort(np.asarray(vals, np.float64))
    n = len(v)
    if n == 0:
        return 0.0, 0.0, 0.0, 0.0
    q = int(math.floor(0.25 * n))
    med = v[(n - 1) // 2] if n % 2 else (v[n // 2] + v[n // 2 - 1]) / 2.0
    lo, hi = (q, n - q) if rej_quartiles else (0, n)
    s = s2 = 0.0
    mad = []
    for i in range(lo, hi):
        s += v[i]
        s2 += v[i] * v[i]
        mad.append(abs(v[i] - med))
    if rej_quartiles:
        n -= 2 * q
    arith = s / n
    mad.sort()
    med_std = 1.4826 * mad[(n - 1) // 2] if n % 2 else 1.4826 * (mad[n // 2] + mad[n // 2 - 1]) / 2.0
    hlp = s2 - n * arith * arith
    std = 0.0 if (round_std and abs(hlp) < 1e-6) else math.sqrt(hlp / (n - 1.0))
    return med, arith, std, med_std


class AutoThEpiOracle:
    """poselib::AutoThEpi (pose_estim.cpp:81-300) over the CPU oracle: `find(th)` -> dict(ok, E, mask) is ARRSAC with refinement."""
    PIX_TH_START, MIN_PIX_TH, MAX_PIX_TH, PIX_MIN_GOOD_TH = 0.5, 0.25, 2.0, 1.6

    def __init__(self, o, f, find):
        self.o, self.f, self.find = o, f, find
        self.cam_th, self.pix_th, self.min_pix_th, self.stable = self.PIX_TH_START * f, self.PIX_TH_START, self.MIN_PIX_TH, False

    def set_corr_th(self, thresh, img=False, store=True):
        pix = thresh if img else thresh / self.f
        pix = self.min_pix_th if pix < self.min_pix_th else (self.MAX_PIX_TH if pix > self.MAX_PIX_TH else pix)
        cam = pix * self.f
        if store:
            self.pix_th, self.cam_th = pix, cam
        return pix if img else cam

    def estimate_thresh(self, a, b, E):
        err = np.sqrt(self.o.get_inliers_strict(a, b, E, 1.0)[2])
        th = self.cam_th
        max_inl = min(4.0 * th, 5.0 * self.f)
        med, arith, std, med_std = full_stats(err, (err.max() - err.min()) > max_inl)
        th_tmp = med + 3.0 * med_std if (arith / med > 2.0 or arith / med < 0.5) else arith + 3.0 * std
        if th_tmp < 5.0 * th or th_tmp < 4.0 * self.PIX_MIN_GOOD_TH:
            return self.set_corr_th(th_tmp, False, False)
        if th < (self.MAX_PIX_TH / 2) * self.f:
            return self.set_corr_th(th * 2.0, False, False)
        return self.set_corr_th(self.min_pix_th, True, False) * self.f

    def estimate_e_var_th(self, a, b, th):
        """-> (rc, E, mask, th, nr_good)"""
        sem = [True, True, False]
        fail_cnt, th_failed = 2, th
        E = mask = None
        nr_good = 0
        while True:
            mask = None
            th_old = th
            r = self.find(th)
            if not r["ok"]:
                if th < self.PIX_MIN_GOOD_TH * self.f and not self.stable and not sem[2]:
                    th_failed = th
                    th = self.PIX_MIN_GOOD_TH * self.f
                    th_old = th
                    r = self.find(th)
                    if not r["ok"]:
                        return -1, None, None, th, nr_good
                    E, mask = r["E"], r["mask"]
                    sem[2] = True
                elif sem[2]:
                    fail_cnt *= 2
                else:
                    return -1, None, None, th, nr_good
            else:
                E, mask = r["E"], r["mask"]
                if sem[2] and fail_cnt > 2:
                    self.min_pix_th = th / self.f
                sem[2] = False
            if fail_cnt <= 2 or not sem[2]:
                nr_good = 0 if mask is None else int(np.count_nonzero(mask))
            if not self.stable:
                th = self.estimate_thresh(a, b, E) if (fail_cnt <= 2 or not sem[2]) else th_failed * fail_cnt
                if sem[2] and th_failed >= th:
                    th = th_failed * fail_cnt
                if not sem[2]:
                    if th_old / th > 1.0 + 1e-6:
                        sem[0] = False
                    elif th_old / th < 1.0 - 1e-6:
                        sem[1] = False
            again = (th_old / th < 0.9 or th / th_old < 0.9) and \
                ((np.float32(nr_good) / np.float32(len(a)) < np.float32(0.67) and (sem[0] or sem[1])) or sem[2])
            if not again:
                break
        return 0, E, mask, th, nr_good


def round_half_away(x):  # std::round
    return math.floor(x + 0.5) if x >= 0 else math.ceil(x - 0.5)


def near_zero(d):
    return -1e-3 < d < 1e-3


def stats(vals):
    """getStatsfromVec(vals, stats, false, false) (pose_helper.cpp:358-413) -> (arithErr, arithStd)."""
    v = np.sort(np.asarray(vals, np.float64))
    n = len(v)
    if n == 0:
        return 0.0, 0.0
    s = s2 = 0.0
    for x in v:
        s += x
        s2 += x * x
    mean = s / n
    hlp = s2 - n * mean * mean
    with np.errstate(invalid="ignore", divide="ignore"):
        std = float(np.sqrt(np.float64(hlp) / (np.float64(n) - 1.0)))
    return mean, std


class StereoRefineOracle:
    def __init__(self, oracle, cfg, K0, K1, dist0, dist1, seed):
        self.o, self.cfg, self.seed = oracle, cfg, seed
        self.K0, self.K1, self.d0, self.d1 = K0, K1, np.asarray(dist0, np.float64), np.asarray(dist1, np.float64)
        self.pix2cam = 4.0 / (math.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))  # :151-153
        self.th = cfg.th_pix_user * self.pix2cam
        self.th2 = self.th * self.th
        self.shrink_log = []
        self.img_size = (800, 600)   # ConfigUSAC's default imgSize (pose_estim.h:112): checkPoolSize indexes an image-sized table with it
        self.check_parameters()
        self.descr_max = F32(0)
        self.resp_max = F32(0)
        self.E = self.R = self.t = None
        self.Q = self.mask_Q = None
        self.mask_E = None
        self.nr_inliers = self.nr_corrs = 0
        self.E_ml = self.R_ml = self.t_ml = None
        self.nr_tries = 0
        self.branch = ""  # which way the last add() went, for the tests
        self.clear()

    def check_parameters(self):  # :187-400
        c = self.cfg
        if c.minStartAggInlRat < 0.075:
            c.minStartAggInlRat = 0.1
        elif c.minStartAggInlRat > 0.75:
            c.minStartAggInlRat = 0.75
        if c.relInlRatThLast > 0.75:
            c.relInlRatThLast = 0.6
        elif c.relInlRatThLast < 0.01:
            c.relInlRatThLast = 0.1
        if c.relInlRatThNew < 0.04:
            c.relInlRatThNew = 0.04
        elif c.relInlRatThNew > 0.55:
            c.relInlRatThNew = 0.35
        if c.minInlierRatSkip > 0.95:
            c.minInlierRatSkip = 0.95
        elif c.minInlierRatSkip < 0.01:
            c.minInlierRatSkip = 0.1
        if c.relMinInlierRatSkip < 0.01:
            c.relMinInlierRatSkip = 0.1
        elif c.relMinInlierRatSkip > 1.0:
            c.relMinInlierRatSkip = 1.0
        if c.maxSkipPairs == 0:
            c.maxSkipPairs = 1
        elif c.maxSkipPairs > 200:
            c.maxSkipPairs = 200
        if c.minInlierRatioReInit <= c.minInlierRatSkip:
            c.minInlierRatioReInit = c.minInlierRatSkip + 0.05
        if c.minInlierRatioReInit > 0.8:
            c.minInlierRatioReInit = 0.8
        elif c.minInlierRatioReInit < 0.15:
            c.minInlierRatioReInit = 0.15
        if c.minPtsDistance < 1.5:
            c.minPtsDistance = 1.5
        if c.minContStablePoses <= 2:
            c.minContStablePoses = 3
        if c.absThRankingStable < 0.01:
            c.absThRankingStable = 0.01
        elif c.absThRankingStable > 0.9:
            c.absThRankingStable = 0.6
        self.max_skip_new = c.maxSkipPairs
        if c.checkPoolPoseRobust != 1 and not c.refineRTold_CorrPool:  # the linear pool solvers are not built (same rule as the library)
            c.checkPoolPoseRobust = 1
        self.check_tmp = c.checkPoolPoseRobust
        self.init_inliers = 0
        self.nr_since_robust = 0
        self.failed_refinements = 0

    def clear(self):  # clearHistoryAndPool :1038-1064
        self.pool = []
        self.next_idx = 0  # corrIdx: only its ratio to the pool size matters (:1233) and only for renumbering, which a list does not need
        self.nr_est = 0
        self.skip = 0
        self.poses, self.rating, self.inl_hist, self.err_hist, self.ml_idx = [], [], [], [], []
        self.max_pool_reached = False
        self.stable = self.ml_stable = False
        self.consec_stable = 0
        self.max_skip_new = self.cfg.maxSkipPairs
        self.q_far = self.q_all = 0

    # ---- primitives ------------------------------------------------------------------------------------------------------------------------
    def inliers(self, E, a, b):
        cnt, mask, err = self.o.get_inliers_strict(a, b, E, self.th2)
        return cnt, mask, err

    def robust(self, a, b):
        """robustPoseEstimation (:1272-1760) on the coordinates (a, b); sets E, R, t, Q, masks, nr_inliers.  False = failed."""
        self.Q = self.mask_Q = None
        method = self.cfg.RobMethod
        n = len(a)
        auto_th = self.cfg.autoTH
        if self.cfg.useRANSAC_fewMatches and n < 100:
            method, auto_th = "RANSAC", False
        if auto_th:  # :1330-1342; th is updated for the frames to come, th2 is not (:159-160)
            if not hasattr(self, "arrsac_rng"):
                self.arrsac_rng = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64)
            auto = AutoThEpiOracle(self.o, self.pix2cam, lambda t: self.o.arrsac_essential(a, b, t, refine=True, rng_state=self.arrsac_rng))
            rc, E, mask, self.th, _ = auto.estimate_e_var_th(a, b, self.th)
            r = dict(ok=(rc == 0), E=E, mask=mask)
        elif method == "RANSAC":
            r = self.o.ransac_essential(a, b, self.th, confidence=0.999, max_iters=1000, lesqu=self.cfg.refineRTold, seed=self.seed)
        elif method == "ARRSAC":
            # the samplers' cv::RNG streams are process-wide in the reference: every estimation continues where the last one stopped
            if not hasattr(self, "arrsac_rng"):
                self.arrsac_rng = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64)
            r = self.o.arrsac_essential(a, b, self.th, refine=self.cfg.refineRTold, rng_state=self.arrsac_rng)
        else:
            r = self.o.lmeds_essential(a, b, confidence=0.999, max_iters=2000, seed=self.seed)
        if not r["ok"]:
            return False
        self.mask_E = r["mask"].copy()
        self.nr_inliers = int(np.count_nonzero(self.mask_E))
        if self.cfg.refineRTold:  # :1460-1474: robustEssentialRefine on the inliers, th / 10
            sel = self.mask_E.astype(bool)
            _, Er, _ = self.o.robust_essential_refine(a[sel], b[sel], r["E"], self.th / 10.0)
            r = dict(r, E=Er)
        good, R, t, Q, m = self.o.recover_pose(r["E"], a, b, self.cfg.maxDist3DPtsZ, r["mask"])
        if good <= 0:
            return False
        self.E, self.R, self.t = r["E"].copy(), R, t / np.sqrt(np.sum(t * t))
        self.Q, self.mask_Q = Q, m
        return True

    # ---- pool ---------------------------------------------------------------------------------------------------------------------------
    def add_to_pool(self, fr):  # addCorrespondencesToPool :1143-1266
        first = len(self.pool) == 0
        errs = []
        for c in self.pool:
            c["age"] += 1
        for i in range(self.nr_corrs):
            if not self.mask_E[i]:
                continue
            c = dict(a=fr["a"][i].copy(), b=fr["b"][i].copy(), age=1, dd=fr["dd"][i], r1=fr["kp1"][i, 2], r2=fr["kp2"][i, 2],
                     pt1=fr["kp1"][i, :2].copy(), pt2=fr["kp2"][i, :2].copy(), errs=[], Q=np.zeros(3), far=False, found=1)
            self.descr_max = max(self.descr_max, c["dd"])
            self.resp_max = max(self.resp_max, c["r1"], c["r2"])
            if first:
                e = self.o.get_inliers_strict(c["a"][None], c["b"][None], self.E, self.th2)[2][0]
                c["errs"].append(e)
                errs.append(e)
                if self.Q is not None:
                    c["Q"] = self.Q[i].copy()
                    c["far"] = not self.mask_Q[i]
                    self.q_all += 1
                    self.q_far += int(c["far"])
            self.pool.append(c)
        if first:
            self.err_hist.append(stats(errs))

    def weight(self, err, dd, r1, r2, far=False, z=0.0):  # computeCorrespondenceWeight :2514-2542
        w_err = 1.0 - err / self.th2
        w_dd = 1.0 - float(dd) / float(self.descr_max)
        w_resp = (float(r1) / float(self.resp_max) + float(r2) / float(self.resp_max)) / 2.0
        w = 0.3 * w_err + 0.5 * w_dd + 0.2 * w_resp
        if far:
            zw = 1.0
            if z > 0:
                zw = 0.5 + 0.9 * self.cfg.maxDist3DPtsZ / (2.0 * z)
            elif z < 0:
                zw = 0.25
            w *= zw
        return w

    def new_is_better(self, new, old):  # compareCorrespondences :2450-2497
        w0 = self.weight(new["err"], new["dd"], new["r1"], new["r2"])
        w1 = self.weight(old["errs"][-1], old["dd"], old["r1"], old["r2"])
        if not w0 > w1:
            rel = (w1 - w0) / w1
            if rel < 0.05 or rel > 0.2:
                return False
        else:
            rel = (w0 - w1) / w0
            if rel < 0.05:
                return False
            if rel > 0.2:
                return True
        if old["age"] > 15:
            return True
        return len(old["errs"]) > 1 and old["errs"][-1] > old["errs"][-2]

    def filter_new(self, fr, err):  # filterNewCorrespondences :2107-2316; returns the filtered frame
        keep = np.flatnonzero(self.mask_E)
        fr = {k: v[keep] for k, v in fr.items()}
        err = err[keep]
        n = len(keep)
        drop_new, drop_old = set(), set()
        if self.pool:
            pts = np.array([c["pt1"] for c in self.pool], F32)
            r2 = F32(self.cfg.minPtsDistance) * F32(self.cfg.minPtsDistance)
            for i in range(n):
                q = fr["kp1"][i, :2]
                dx, dy = pts[:, 0] - q[0], pts[:, 1] - q[1]
                d2 = dx * dx + dy * dy  # float32, like the kd-tree's metric
                near = np.flatnonzero(d2 < r2)
                if len(near) == 0:
                    continue
                near = near[np.lexsort((near, d2[near]))]
                new = dict(err=err[i], dd=fr["dd"][i], r1=fr["kp1"][i, 2], r2=fr["kp2"][i, 2])
                marked = False
                j = 0
                while j < len(near):
                    old = self.pool[near[j]]
                    if not d2[near[j]] < F32(2.0):
                        break
                    ex, ey = old["pt2"][0] - fr["kp2"][i, 0], old["pt2"][1] - fr["kp2"][i, 1]
                    dd2 = float(ex) * float(ex) + float(ey) * float(ey)
                    if dd2 < 2.0:
                        if dd2 < 0.01 and d2[near[j]] < F32(0.01):
                            drop_new.add(i)
                            marked = True
                            old["found"] += 1
                            break
                        if self.new_is_better(new, old):
                            drop_old.add(int(near[j]))
                        else:
                            drop_new.add(i)
                            marked = True
                            old["found"] += 1
                            break
                    j += 1
                if not marked and j == 0:
                    while j < len(near):
                        if not self.new_is_better(new, self.pool[near[j]]):
                            drop_new.add(i)
                            break
                        j += 1
                    if j >= len(near):
                        drop_old.update(int(x) for x in near)
        stay = np.array([i for i in range(n) if i not in drop_new], np.int64)
        fr = {k: v[stay] for k, v in fr.items()}
        self.nr_corrs = self.nr_inliers = len(stay)
        self.mask_E = np.ones(len(stay), np.uint8)
        self.Q = self.mask_Q = None
        self.delete_from_pool(sorted(drop_old))
        return fr

    def delete_from_pool(self, positions):  # poolCorrespondenceDelete :2318-2436
        dead = set(positions)
        if not dead:
            return
        for p in dead:
            c = self.pool[p]
            if not near_zero(100.0 * float(c["Q"].sum())) and self.q_all:
                self.q_all -= 1
                if c["far"] and self.q_far:
                    self.q_far -= 1
        self.pool = [c for p, c in enumerate(self.pool) if p not in dead]
        if not self.pool:
            self.q_all = self.q_far = 0

    def shrink_pool(self, max_size):  # checkPoolSize :2550-2816
        """Too many correspondences: thin the pool where the left image is densely covered.  (1) correspondences that share a rounded left
        pixel with another one: all but the best weight of a pixel go -- or, when there are more of them than the quota, the reference's loop
        over a weight list that starts with len(idx1) value-initialised (0.0, 0) entries; (2) density image of the occupied pixels, dilated with
        the elliptic element of minPtsDistance and eroded with the next larger one, AND-ed with the occupied pixels: what is left lies inside
        dense regions and goes -- everything while fewer than the quota (then the element grows), else the lowest weights among them.
        An independent restatement (numpy shifts; OpenCV's element / anchor / border conventions from their published definitions)."""
        n = len(self.pool)
        if max_size < 0 and n > 20:
            n_del = n // 2
        elif max_size < 0 or n <= max_size:
            return
        else:
            n_del = n - max_size
        if n - n_del < 10:
            if n > 20:
                n_del = n // 2
            else:
                return
        W, H = self.img_size
        wgt = lambda i: self.weight(self.pool[i]["errs"][-1], self.pool[i]["dd"], self.pool[i]["r1"], self.pool[i]["r2"], self.pool[i]["far"],  # noqa: E731
                                    self.pool[i]["Q"][2])
        pix = [(int(round_half_away(float(c["pt1"][0]))), int(round_half_away(float(c["pt1"][1])))) for c in self.pool]   # (x, y)
        cells, multi = {}, []
        for i, xy in enumerate(pix):
            assert 0 <= xy[0] < W and 0 <= xy[1] < H
            cells.setdefault(xy, []).append(i)
            if len(cells[xy]) == 2:
                multi.append(xy)
        dele = []
        rounds_done, by_weight = 0, False
        if multi:
            idx1 = [(m, j) for m, xy in enumerate(multi) for j in range(len(cells[xy]))]
            nr = [len(cells[xy]) for xy in multi]
            n_multi = len(idx1) - len(multi)
            if n_multi <= n_del:
                for xy in multi:
                    order = sorted(range(len(cells[xy])), key=lambda j: -wgt(cells[xy][j]))
                    dele += [cells[xy][j] for j in order[1:]]
                    cells[xy] = [cells[xy][order[0]]]
                n_del -= n_multi
            else:
                lst = [(0.0, 0)] * len(idx1) + [(wgt(cells[multi[m]][j]), k) for k, (m, j) in enumerate(idx1)]
                lst.sort(key=lambda t: t[0])
                count = 0
                for _, k in lst:
                    m, j = idx1[k]
                    if nr[m] > 1:
                        dele.append(cells[multi[m]][j])
                        nr[m] -= 1
                        count += 1
                    if count >= n_del:
                        break
                n_del = 0
        if n_del:
            dens = np.zeros((H, W), bool)
            for x, y in pix:
                dens[y, x] = True
            init = dens.copy()

            def ellipse(size):
                k = np.zeros((size, size), bool)
                r = c = size // 2
                for i in range(size):
                    dy = i - r
                    if abs(dy) <= r:
                        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / (r * r)))) if r else 0
                        k[i, max(c - dx, 0):min(c + dx + 1, size)] = True
                return k

            def shifted(img, dy, dx):   # out[y, x] = img[y + dy, x + dx], False outside
                out = np.zeros_like(img)
                ys, xs = slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx))
                yd, xd = slice(max(0, dy), H - max(0, -dy)), slice(max(0, dx), W - max(0, -dx))
                out[ys, xs] = img[yd, xd]
                return out

            def morph(img, k, dilate):
                a = k.shape[0] // 2
                acc = np.zeros_like(img) if dilate else np.ones_like(img)
                for i, j in zip(*np.nonzero(k)):
                    sh = shifted(img, i - a, j - a)
                    acc = (acc | sh) if dilate else (acc & sh)
                return acc

            mpd = float(self.cfg.minPtsDistance)
            size = int(np.ceil(mpd)) + 1 if near_zero(mpd - np.ceil(mpd)) else int(np.ceil(mpd))
            rounds = 0
            while n_del:
                dens = morph(morph(dens, ellipse(size), True), ellipse(size + 1), False) & init
                loc = [(x, y) for y, x in zip(*np.nonzero(dens))]      # row-major, as cv::findNonZero
                rounds_done += 1
                if len(loc) <= n_del and rounds < 100:
                    dele += [cells[xy][0] for xy in loc]
                    n_del -= len(loc)
                    if n_del:
                        dens = ~dens & init
                        init = dens.copy()
                        size += 1
                        rounds += 1
                else:
                    if rounds >= 100:
                        loc = [(x, y) for y, x in zip(*np.nonzero(init))]
                    by_weight = True
                    order = sorted(range(len(loc)), key=lambda i: wgt(cells[loc[i]][0]))
                    dele += [cells[loc[i]][0] for i in order[:n_del]]
                    n_del = 0
        self.shrink_log.append(dict(deleted=len(dele), shared_pixels=len(multi), density_rounds=rounds_done, by_weight=by_weight))
        self.delete_from_pool(dele)
        self.max_pool_reached = True

    # ---- rating / stability ---------------------------------------------------------------------------------------------------------------
    def near_to_mean(self):  # getNearToMeanPose :2817-3129
        n_p = len(self.poses)
        if n_p < 5:
            return -1
        point = np.array([0.5, 0.5, 0.5])
        res = np.array([[sum(R[r, k] * point[k] for k in range(3)) + t[r] for r in range(3)] for (_, R, t) in self.poses])
        srt = [sorted(range(n_p), key=lambda i, a=a: res[i, a]) for a in range(3)]  # stable, ascending
        val = [[res[i, a] for i in srt[a]] for a in range(3)]
        q0 = int(math.floor(n_p * 0.25 + 0.5))
        q1 = n_p - q0
        over = any(abs(val[a][-1] - val[a][0]) > 0.05 for a in range(3))
        med = [val[a][(n_p - 1) // 2] if n_p % 2 else (val[a][n_p // 2] + val[a][n_p // 2 - 1]) / 2.0 for a in range(3)]
        nq = n_p - 2 * q0
        lo, hi, mean_all = [0.0] * 3, [0.0] * 3, [0.0] * 3
        for a in range(3):
            su = sum(val[a][:q0])
            sm = sum(val[a][q0:q1])
            so = sum(val[a][q1:])
            cu = sum(x * x for x in val[a][:q0])
            cm = sum(x * x for x in val[a][q0:q1])
            co = sum(x * x for x in val[a][q1:])
            mean_all[a] = (so + (su + sm)) / n_p
            mean_mid = sm / nq
            with np.errstate(invalid="ignore", divide="ignore"):
                if over:
                    sd = float(np.sqrt(np.float64(cm - nq * mean_mid * mean_mid) / (np.float64(nq) - 1.0)))
                    lo[a], hi[a] = mean_mid - 3.0 * sd, mean_mid + 3.0 * sd
                else:
                    c2 = cm + (cu + co)
                    sd = float(np.sqrt(np.float64(c2 - n_p * mean_all[a] * mean_all[a]) / (np.float64(n_p) - 1.0)))
                    lo[a], hi[a] = mean_all[a] - 3.0 * sd, mean_all[a] + 3.0 * sd
        possible = [True] * 3
        for a in range(3):
            m, d = mean_all[a], med[a]
            if (m > 0 and d > 0) or (m < 0 and d < 0):
                if m / d > 1.33 or d / m > 1.33 or abs(m - d) > 0.02:
                    possible[a] = False
            elif near_zero(m) or near_zero(d):
                if abs(m - d) > 0.02:
                    possible[a] = False
            else:
                possible[a] = False
        valid = []
        if not any(possible):
            mid = [srt[a][q0:q1] for a in range(3)]
            for v in mid[0]:
                if v in mid[1] and v in mid[2]:
                    valid.append(v)
        else:
            idx = []
            for a in range(3):
                if possible[a]:
                    idx.append([srt[a][i] for i in range(n_p) if lo[a] < val[a][i] < hi[a]])
                else:
                    idx.append(srt[a][q0:q1])
            for i, v in enumerate(idx[0]):
                if v in idx[1] and v in idx[2]:
                    valid.append(srt[0][i])  # the reference takes the i-th entry of the sorted x list here (:3037), not v
        if len(valid) < 3:
            return -2
        cg = np.zeros(3)
        for v in valid:
            cg += res[v]
        cg /= len(valid)
        dist = np.sqrt(((res - cg) ** 2).sum(axis=1))
        imin = int(np.argmin(dist))
        imax = n_p - 1 - int(np.argmax(dist[::-1]))
        self.E_ml, self.R_ml, self.t_ml = self.poses[imin]
        self.ml_idx.append(imin)
        max_dist = dist[imax] + np.sqrt((cg * cg).sum()) * 0.0075
        self.rating = [1.0 - d / max_dist for d in dist]
        return 0

    def update_max_skip(self):  # :3300-3317
        r = self.cfg.raiseSkipCnt
        if (r & 0xF) and (((r & 0xF0) >> 4) + 1) <= self.consec_stable:
            self.max_skip_new = int(math.ceil(self.cfg.maxSkipPairs * (1.0 + (r & 0xF) * 0.25)))
        else:
            self.max_skip_new = self.cfg.maxSkipPairs

    def check_stability(self):  # checkPoseStability :3131-3298
        c = self.cfg
        err = self.near_to_mean()
        if err:
            self.stable = self.ml_stable = False
            self.E_ml, self.R_ml, self.t_ml = self.E, self.R, self.t
            if err != -2:
                self.nr_tries = 0
            return
        if self.nr_est < c.minContStablePoses or len(self.pool) < 1000:
            self.stable = self.ml_stable = False
            self.nr_tries = 0
            return
        last = self.rating[-1]
        count = stable = 2
        while count <= c.minContStablePoses:
            r = self.rating[self.nr_est - count]
            if last - c.absThRankingStable < r < last + c.absThRankingStable and r > c.minNormDistStable:
                stable += 1
            else:
                stable -= 1
                break
            count += 1
        if len(self.ml_idx) >= c.minContStablePoses:
            last_idx = self.ml_idx[-1]
            if self.rating[last_idx] > c.minNormDistStable:
                self.ml_stable = all(i == last_idx for i in self.ml_idx[-c.minContStablePoses:-1])
            else:
                self.ml_stable = False
        far = self.q_far / self.q_all if self.q_all else 0.0
        if stable == count and far < 0.95:
            self.stable = True
            self.consec_stable += 1
            if self.max_skip_new <= c.maxSkipPairs:
                self.update_max_skip()
            if self.nr_tries:
                self.nr_tries -= 1
            return
        self.stable = False
        self.nr_tries += 1
        if self.nr_tries > c.minContStablePoses and self.max_pool_reached and far < c.maxRat3DPtsFar:
            m = c.minContStablePoses
            rng = [(self.err_hist[self.nr_est - k - 1][0] - 2.0 * self.err_hist[self.nr_est - k - 1][1],
                    self.err_hist[self.nr_est - k - 1][0] + 2.0 * self.err_hist[self.nr_est - k - 1][1]) for k in range(m)]
            mean = sum(self.err_hist[self.nr_est - k - 1][0] for k in range(m)) / m
            min_lo, max_lo = min(r[0] for r in rng), max(r[0] for r in rng)
            min_hi, max_hi = min(r[1] for r in rng), max(r[1] for r in rng)
            if min_hi <= min_lo or max_lo >= max_hi:
                self.stable = False
                self.consec_stable = 0
                return
            e0, e1 = mean - min_lo, max_hi - mean
            p0, p1 = e0 / (e0 + e1), e1 / (e0 + e1)
            for r in rng:
                if p1 * (r[1] - mean) / e1 + p0 * (mean - r[0]) / e0 < 0.8:
                    self.stable = False
                    self.consec_stable = 0
                    return
            self.stable = True
            self.consec_stable += 1
        else:
            self.consec_stable = 0
        if self.stable and self.max_skip_new <= c.maxSkipPairs:
            self.update_max_skip()

    # ---- one image pair (:416-957) ----------------------------------------------------------------------------------------------------------
    def init_after_reinit(self, ratio, fr):
        self.add_to_pool(fr)
        self.poses.append((self.E.copy(), self.R.copy(), self.t.copy()))
        self.inl_hist.append(ratio)
        self.nr_est += 1

    def reinit(self, ratio, fr):
        self.clear()
        self.init_after_reinit(ratio, fr)

    def pool_coords(self):
        return np.array([c["a"] for c in self.pool]), np.array([c["b"] for c in self.pool])

    def robust_on_pool(self):
        a, b = self.pool_coords()
        return self.robust(a, b)

    def refine_from_pool(self):
        """refinePoseFromPool (:1767-2084) for refineRTold_CorrPool: robustEssentialRefine of the current E on all pool correspondences,
        R, t, Q from the refined matrix, E rebuilt from R and t (getEfromRT, pose_helper.cpp:785-788)."""
        a, b = self.pool_coords()
        n = len(a)
        self.nr_inliers = n
        self.Q = self.mask_Q = None
        _, E, _ = self.o.robust_essential_refine(a, b, self.E, self.th / 10.0)
        self.mask_E = np.ones(n, np.uint8)
        good, R, t, Q, m = self.o.recover_pose(E, a, b, self.cfg.maxDist3DPtsZ, np.ones(n, np.uint8))
        if good <= 0:
            return False
        t = t / np.sqrt(np.sum(t * t))
        tv = np.asarray(t).reshape(-1)
        S = np.array([[0, -tv[2], tv[1]], [tv[2], 0, -tv[0]], [-tv[1], tv[0], 0]])
        self.E, self.R, self.t = S @ R, R, t
        self.Q, self.mask_Q = Q, m
        return True

    def add(self, kp1, kp2, dd):
        """kp1 / kp2: n x 3 float32 (x, y, response); dd: n float32 descriptor distances.  Returns the reference's return code."""
        c = self.cfg
        self.nr_corrs = len(dd)
        a = self.o.img_to_cam(kp1[:, :2], self.K0)
        b = self.o.img_to_cam(kp2[:, :2], self.K1)
        ok, a, b = self.o.remove_lens_dist(a, b, self.d0, self.d1)
        if not ok:
            return -1
        fr = dict(a=a.astype(np.float64), b=b.astype(np.float64), kp1=kp1, kp2=kp2, dd=dd)
        a_all, b_all = fr["a"], fr["b"]
        self.branch = ""
        if self.nr_est == 0:
            self.branch = "init"
            self.check_tmp = c.checkPoolPoseRobust  # robustInitialization :973-978
            if not self.robust(fr["a"], fr["b"]):
                return -1
            self.init_inliers = self.nr_inliers
            ratio = self.nr_inliers / self.nr_corrs
            if ratio < c.minStartAggInlRat:
                return 0  # err -3 of robustInitialization -> 0
            self.init_after_reinit(ratio, fr)
            return 0
        cnt, mask, err = self.inliers(self.E, fr["a"], fr["b"])
        ratio = cnt / self.nr_corrs
        ratio1 = 0.0
        to_pool = False
        if ratio < (1.0 - c.relInlRatThLast) * self.inl_hist[-1]:
            if not self.robust(fr["a"], fr["b"]):
                return -1
            ratio1 = self.nr_inliers / self.nr_corrs
            if ratio < ratio1 * (1.0 - c.relInlRatThNew):
                if ratio1 >= c.minInlierRatioReInit and ratio < c.minInlierRatioReInit:
                    self.reinit(ratio1, fr)
                    self.branch = "pose_changed"
                    return 0
                if ratio1 < c.minInlierRatSkip and ratio1 < c.relMinInlierRatSkip * self.inl_hist[-1]:
                    self.E, self.R, self.t = (x.copy() for x in self.poses[-1])
                    self.branch = "restore_last"
                else:
                    self.branch = "pool_only"
                    saved = (self.E, self.R, self.t, self.mask_E, self.mask_Q, self.Q, self.nr_inliers)
                    if not self.robust_on_pool():
                        (self.E, self.R, self.t, self.mask_E, self.mask_Q, self.Q, self.nr_inliers) = saved
                        self.reinit(ratio1, fr)
                        return 0
                    (_, _, _, self.mask_E, self.mask_Q, self.Q, self.nr_inliers) = saved
                    self.stable = self.ml_stable = False
                self.skip += 1
            else:
                self.branch = "low_ratio_pool"
                self.mask_Q = self.Q = None
                self.mask_E = mask
                self.nr_inliers = cnt
                ratio1 = ratio
                self.E, self.R, self.t = (x.copy() for x in self.poses[-1])
                to_pool = True
        else:
            self.branch = "pool"
            to_pool = True
            self.mask_E = mask
            self.nr_inliers = cnt
            ratio1 = ratio
        if to_pool:
            fr = self.filter_new(fr, err)
            n_new = len(fr["dd"])
            if n_new + len(self.pool) > c.maxPoolCorrespondences:
                self.shrink_pool(c.maxPoolCorrespondences - n_new)
                self.branch += "+shrink"
            n_before = len(self.pool)
            self.add_to_pool(fr)
            old = (self.E, self.R, self.t)
            saved = (self.mask_E, self.nr_inliers)
            min_rel = 0.75
            # robust estimation on the pool or refinement of the last pose on it (:680-820)
            if c.checkPoolPoseRobust == 1 or self.nr_since_robust > self.check_tmp or \
                    (not self.max_pool_reached and self.check_tmp * self.init_inliers < len(self.pool)):
                self.mask_Q = self.Q = None
                if not self.robust_on_pool():
                    self.E, self.R, self.t = old
                    self.mask_E, self.nr_inliers = saved
                    self.Q = None
                    self.reinit(ratio1, fr)
                    return -3
                if c.checkPoolPoseRobust > 1:
                    if self.max_pool_reached:
                        self.check_tmp = c.checkPoolPoseRobust if c.checkPoolPoseRobust > 10 else 10
                    elif self.check_tmp > 50:
                        self.check_tmp = int(c.maxPoolCorrespondences) // self.init_inliers + 2
                    else:
                        self.check_tmp = int(round_half_away(c.checkPoolPoseRobust + math.exp(0.8 + self.check_tmp / 6.0)))
                self.nr_since_robust = 0
                min_rel = 0.7
                if c.checkPoolPoseRobust != 1:
                    self.branch += "+robust"
            else:
                self.nr_since_robust = self.nr_since_robust + 1 if self.max_pool_reached else 0
                self.branch += "+refined"
                if not self.refine_from_pool():
                    self.E, self.R, self.t = old
                    self.skip += 1
                    if self.failed_refinements > 0:
                        self.failed_refinements = 0
                        self.clear()
                    else:
                        self.delete_from_pool(list(range(n_before, len(self.pool))))
                        self.failed_refinements += 1
                    return -3
                self.failed_refinements = 0
            if self.nr_inliers < min_rel * len(self.pool):
                self.E, self.R, self.t = old
                self.clear()
                self.branch += "+pool_lost"
                return -3
            cnt, mask, err = self.inliers(self.E, a_all, b_all)
            ratio = cnt / len(a_all)
            if ratio < ratio1 * (1 - c.relInlRatThNew):
                self.E, self.R, self.t = old
                self.clear()
                self.branch += "+pair_lost"
                return -3
            self.inl_hist.append(ratio)
            self.poses.append((self.E.copy(), self.R.copy(), self.t.copy()))
            e_in = np.zeros(cnt)
            k = 0
            for i in range(cnt):  # only the first `cnt` entries of the mask are walked (:836-841)
                if mask[i]:
                    e_in[k] = err[i]
                    k += 1
            self.err_hist.append(stats(e_in))
            if self.nr_inliers < len(self.pool):
                keep = np.flatnonzero(self.mask_E)
                self.delete_from_pool([int(p) for p in np.flatnonzero(self.mask_E == 0)])
                self.Q, self.mask_Q = self.Q[keep], self.mask_Q[keep]
            pa, pb = self.pool_coords()
            e = self.o.get_inliers_strict(pa, pb, self.E, self.th2)[2] if len(pa) else []
            for k, cpt in enumerate(self.pool):
                if self.Q is not None:
                    cpt["Q"] = self.Q[k].copy()
                    cpt["far"] = not self.mask_Q[k]
                    if cpt["age"] <= 1:
                        self.q_all += 1
                        self.q_far += int(cpt["far"])
                cpt["errs"].append(e[k])
            self.nr_est += 1
            self.skip = 0
            self.check_stability()
        if self.skip > self.max_skip_new:
            self.reinit(ratio1, fr)
            self.branch += "+skip_reinit"
        return 0
