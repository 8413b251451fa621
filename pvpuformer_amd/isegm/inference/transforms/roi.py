"""Region-of-interest arithmetic of the zoom transform as one value type.  A ``Roi`` is the inclusive pixel box
``(r0, r1, c0, c1)`` the reference keeps as a bare tuple (it compares equal to, and unpacks like, that tuple, so predictor
states stay interchangeable).  The arithmetic -- Python floats, ``round`` to even, the ``+ 1`` of inclusive extents -- is
what pins the crops to the reference's (tests/golden/zoom.npz; isegm/utils/misc.py:36-79, zoom_in.py:153-196)."""


class Roi(tuple):
    __slots__ = ()

    def __new__(cls, r0, r1, c0, c1):
        return super().__new__(cls, (int(r0), int(r1), int(c0), int(c1)))

    @classmethod
    def whole(cls, height, width):
        return cls(0, height - 1, 0, width - 1)

    @property
    def height(self):
        return self[1] - self[0] + 1

    @property
    def width(self):
        return self[3] - self[2] + 1

    def grown(self, ratio, floor=None):
        """Scaled about its centre by ``ratio``, each side at least ``floor`` pixels long."""
        out = []
        for lo, hi in ((self[0], self[1]), (self[2], self[3])):
            mid, side = 0.5 * (lo + hi), ratio * (hi - lo + 1)
            if floor is not None:
                side = max(side, floor)
            out += [int(round(mid - 0.5 * side)), int(round(mid + 0.5 * side))]
        return Roi(*out)

    def clipped(self, height, width):
        return Roi(max(0, self[0]), min(height - 1, self[1]), max(0, self[2]), min(width - 1, self[3]))

    def overlap(self, other):
        """Product of the two 1-D intersection-over-union ratios (inclusive extents; the quantity the re-crop rule thresholds)."""
        ratio = 1.0
        for k in (0, 2):
            inter = min(self[k + 1], other[k + 1]) - max(self[k], other[k]) + 1
            span = max(self[k + 1], other[k + 1]) - min(self[k], other[k]) + 1
            ratio *= max(0, inter) / max(1e-6, span)
        return ratio

    def encloses(self, clicks):
        """Every positive click lies in [r0, r1) x [c0, c1) -- the upper bounds are exclusive in this check (zoom_in.py:188-196)."""
        return all(self[0] <= c.coords[0] < self[1] and self[2] <= c.coords[1] < self[3] for c in clicks if c.is_positive)

    def to_crop(self, coords, crop_hw):
        """Image coordinates -> coordinates in the crop resized to ``crop_hw``."""
        return (crop_hw[0] * (coords[0] - self[0]) / self.height, crop_hw[1] * (coords[1] - self[2]) / self.width)

    def output_size(self, target):
        """A (height, width) target is taken as is; a scalar is the length of the longer side."""
        if isinstance(target, tuple):
            return target
        scale = target / max(self.height, self.width)
        return int(round(self.height * scale)), int(round(self.width * scale))
