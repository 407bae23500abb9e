"""TEST INFRASTRUCTURE (only tests/ may import this): the occupancy-bitmap rule build of csrc/scn_rules.hip (scn_occ_plan,
k_occ_set, k_occ_coarsen, k_occ_count / k_occ_base, k_occ_sites, k_occ_rows, k_sub_probe_occ, the coarse lookup of
k_str_fill_ml) restated in numpy, word for word -- 32 occupancy bits + a running count per word, the same
box, the same index arithmetic -- so that a CPU test can hold the ALGORITHM against oracle/rules.py, which restates the reference
(site order = rank of the voxel key per sample, sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:72-119,210-229 and
extra/cudpp/src/cudpp_hash/hash_multivalue.cpp:45-110; the 27 neighbour queries sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:45-114;
the 2/2 coarse site sparseconvnet/SCN/Metadata/ConvolutionRules.h:344-378).  The GPU tests hold the HIP kernels against the same
oracle bit for bit; this file says WHY they agree: a rank in key order (sample, z, y, x) is a prefix sum of population counts over
a bitmap laid out in that order."""
import numpy as np


class Box(object):
    """scn_occ_plan: origin aligned to 2^(levels-1) (so that scale l is scale 0 shifted by l, relative coordinates included),
    extents per scale, words per x row, first word of every scale"""

    def __init__(self, coords, levels):
        c = np.asarray(coords, np.int64)
        if c.shape[1] == 3:
            c = np.concatenate([c, np.zeros((len(c), 1), np.int64)], 1)
        self.levels = levels
        lo, hi = c.min(0), c.max(0)
        align = 1 << (levels - 1)
        self.origin = np.array([lo[0] & ~(align - 1), lo[1] & ~(align - 1), lo[2] & ~(align - 1), lo[3]], np.int64)
        self.nb = int(hi[3] - lo[3] + 1)
        self.n = [[int(((hi[a] - self.origin[a]) >> l) + 1) for a in range(3)] for l in range(levels)]
        self.wx = [(n[0] + 31) // 32 for n in self.n]
        self.woff = [0]
        for l in range(levels):
            self.woff.append(self.woff[-1] + self.nb * self.n[l][2] * self.n[l][1] * self.wx[l])

    def word(self, l, rx, ry, rz, rb):
        """occ_word_index: the linear order of the words is the key order (sample, z, y, x)"""
        nx, ny, nz = self.n[l]
        return self.woff[l] + ((rb * nz + rz) * ny + ry) * self.wx[l] + (rx >> 5)


class Occupancy(object):
    def __init__(self, coords, levels):
        c = np.asarray(coords, np.int64)
        if c.shape[1] == 3:
            c = np.concatenate([c, np.zeros((len(c), 1), np.int64)], 1)
        self.coords = c
        self.box = b = Box(c, levels)
        W = b.woff[levels]
        self.bits = np.zeros(W, np.uint32)
        # k_occ_set: one OR per input row at scale 0
        r = c - b.origin
        w = b.word(0, r[:, 0], r[:, 1], r[:, 2], r[:, 3])
        np.bitwise_or.at(self.bits, w, (np.uint32(1) << (r[:, 0] & 31).astype(np.uint32)))
        # k_occ_coarsen: a coarse word = two fine words of each of 2 x 2 rows, OR-ed, neighbouring bits OR-ed, even bits kept
        for l in range(levels - 1):
            nxc, nyc, nzc = b.n[l + 1]
            nxf, nyf, nzf = b.n[l]
            for rb in range(b.nb):
                for zc in range(nzc):
                    for yc in range(nyc):
                        for xw in range(b.wx[l + 1]):
                            f = 0
                            for dz in (0, 1):
                                for dy in (0, 1):
                                    yf, zf = 2 * yc + dy, 2 * zc + dz
                                    if yf < nyf and zf < nzf:
                                        row = b.woff[l] + ((rb * nzf + zf) * nyf + yf) * b.wx[l]
                                        f |= int(self.bits[row + 2 * xw])
                                        if 2 * xw + 1 < b.wx[l]:
                                            f |= int(self.bits[row + 2 * xw + 1]) << 32
                            t = f | (f >> 1)
                            out = 0
                            for i in range(32):
                                out |= ((t >> (2 * i)) & 1) << i
                            self.bits[b.woff[l + 1] + ((rb * nzc + zc) * nyc + yc) * b.wx[l + 1] + xw] = out
        # k_occ_count / k_occ_base: the running count in front of every word, counted through all scales
        pop = np.array([bin(int(v)).count("1") for v in self.bits], np.int64)
        self.base = np.concatenate([[0], np.cumsum(pop)[:-1]])
        self.start = [int(self.base[b.woff[l]]) if b.woff[l] < W else int(pop.sum()) for l in range(levels)] + [int(pop.sum())]
        self.n_sites = [self.start[l + 1] - self.start[l] for l in range(levels)]

    def rank(self, l, x, y, z, b):
        """site of voxel (x, y, z, b) at scale l or -1: one word, one population count (occ_rank; the probe of k_sub_probe_occ)"""
        bx = self.box
        rx, ry, rz, rb = x - (int(bx.origin[0]) >> l), y - (int(bx.origin[1]) >> l), z - (int(bx.origin[2]) >> l), b - int(bx.origin[3])
        nx, ny, nz = bx.n[l]
        if not (0 <= rx < nx and 0 <= ry < ny and 0 <= rz < nz and 0 <= rb < bx.nb):
            return -1
        w = bx.word(l, rx, ry, rz, rb)
        bits = int(self.bits[w])
        if not (bits >> (rx & 31)) & 1:
            return -1
        return int(self.base[w]) + bin(bits & ((1 << (rx & 31)) - 1)).count("1") - self.start[l]

    def sites(self, l):
        """k_occ_sites: coordinates of the sites of scale l in rank order"""
        bx = self.box
        nx, ny, nz = bx.n[l]
        out = []
        for w in range(bx.woff[l], bx.woff[l + 1]):
            bits = int(self.bits[w])
            if not bits:
                continue
            wi = w - bx.woff[l]
            xw, row = wi % bx.wx[l], wi // bx.wx[l]
            ry, rz, rb = row % ny, (row // ny) % nz, row // (ny * nz)
            for i in range(32):
                if (bits >> i) & 1:
                    out.append(((int(bx.origin[0]) >> l) + xw * 32 + i, (int(bx.origin[1]) >> l) + ry, (int(bx.origin[2]) >> l) + rz,
                                int(bx.origin[3]) + rb))
        return np.array(out, np.int64).reshape(-1, 4)

    def site_of_row(self):
        """k_occ_rows: the site of every input row"""
        c = self.coords
        return np.array([self.rank(0, int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in c], np.int64)

    def neighbours(self, l):
        """k_sub_probe_occ: int32 [N, 27], k = (dx+1)*9 + (dy+1)*3 + (dz+1)"""
        s = self.sites(l)
        nbr = np.full((len(s), 27), -1, np.int32)
        for o, (x, y, z, b) in enumerate(s.tolist()):
            k = 0
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        nbr[o, k] = self.rank(l, x + dx, y + dy, z + dz, b)
                        k += 1
        return nbr

    def coarse_of_fine(self, l):
        """the lookup of k_str_fill_ml: the site of scale l + 1 that holds a site of scale l"""
        s = self.sites(l)
        return np.array([self.rank(l + 1, x >> 1, y >> 1, z >> 1, b) for x, y, z, b in s.tolist()], np.int64)
