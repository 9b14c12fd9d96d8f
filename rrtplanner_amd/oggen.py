"""Seeded fractal-noise occupancy grids (workload generator).

Counterpart of the reference's ``perlin_occupancygrid`` (/root/reference/rrtplanner/oggen.py:7-45),
which calls the third-party ``pyfastnoisesimd`` (pinned ==0.4.2 in the reference's
requirements.txt:24, not vendored, unseeded).  The arithmetic of that package is not
reproducible even inside the reference, so parity for grid *values* is unpinned
(SURVEY.md 8(c)); what is kept is the contract of the function: fractal gradient noise,
min-max normalised to [0,1], ``np.where(noise >= thresh, 0, 1)`` -> int64 ``(w,h)`` grid (or
``(frames,w,h)``), 1 = obstacle, 0 = free.

This generator is our own design: multi-octave lattice gradient noise with quintic
fade, evaluated with numpy on the host, fully determined by ``seed``.
"""
import numpy as np

__all__ = ["perlin_occupancygrid", "noise_field", "noise_lattices", "DeviceGrids", "largest_free_component", "random_connected_pair"]


def _fade(t):
    return t * t * t * (t * (t * 6.0 - 15.0) + 10.0)


def noise_lattices(w: int, h: int, frames: int = None, seed: int = 0, base_cell: float = None, octaves: int = 4):
    """The seeded gradient lattices of the noise field: list of (cell edge in pixels, amplitude, unit gradients
    (nz, nx, ny, 3)) per octave.  Host and device generators evaluate the same lattices."""
    rng = np.random.default_rng(seed)
    f = 1 if frames is None else frames
    cell = base_cell if base_cell is not None else max(8.0, min(w, h) / 6.0)
    amp = 1.0
    out = []
    for _ in range(octaves):
        nz, nx, ny = int(np.ceil(f / cell)) + 2, int(np.ceil(w / cell)) + 2, int(np.ceil(h / cell)) + 2
        g = rng.normal(size=(nz, nx, ny, 3))
        g /= np.linalg.norm(g, axis=-1, keepdims=True)
        out.append((float(cell), float(amp), g))
        amp *= 0.5
        cell = max(2.0, cell / 2.0)
    return out


def _gradient_noise3(shape, cell, g):
    """One octave of 3-D gradient noise on an integer lattice of spacing `cell` (float64)."""
    f, w, h = shape
    z = np.arange(f) / cell
    x = np.arange(w) / cell
    y = np.arange(h) / cell
    z0, x0, y0 = np.floor(z).astype(int), np.floor(x).astype(int), np.floor(y).astype(int)
    fz, fx, fy = (z - z0)[:, None, None], (x - x0)[None, :, None], (y - y0)[None, None, :]
    uz, ux, uy = _fade(fz), _fade(fx), _fade(fy)
    out = np.zeros(shape)
    for dz in (0, 1):
        wz = uz if dz else 1.0 - uz
        for dx in (0, 1):
            wx = ux if dx else 1.0 - ux
            for dy in (0, 1):
                wy = uy if dy else 1.0 - uy
                gg = g[(z0 + dz)[:, None, None], (x0 + dx)[None, :, None], (y0 + dy)[None, None, :]]
                dot = gg[..., 0] * (fz - dz) + gg[..., 1] * (fx - dx) + gg[..., 2] * (fy - dy)
                out += wz * wx * wy * dot
    return out


def noise_field(w: int, h: int, frames: int = None, seed: int = 0, base_cell: float = None, octaves: int = 4):
    """Fractal gradient noise, float32, shape (w,h) or (frames,w,h)."""
    f = 1 if frames is None else frames
    acc = np.zeros((f, w, h))
    for cell, amp, g in noise_lattices(w, h, frames, seed, base_cell, octaves):
        acc += amp * _gradient_noise3((f, w, h), cell, g)
    acc = acc.astype(np.float32)
    return acc[0] if frames is None else acc


def perlin_occupancygrid(w: int, h: int, thresh: float = 0.33, frames: int = None, seed: int = 1) -> np.ndarray:
    """Same signature as the reference (oggen.py:7-9) plus ``seed``.

    Returns an int64 grid, 1 = obstacle, 0 = free (oggen.py:40-45).
    """
    xynoise = noise_field(w, h, frames=frames, seed=seed)
    xynoise = xynoise - xynoise.min()
    xynoise = xynoise / (xynoise.max() - xynoise.min())
    return np.where(xynoise >= thresh, 0, 1)


def largest_free_component(og: np.ndarray) -> np.ndarray:
    """Boolean mask of the largest 8-connected free region (start/goal must share one:
    the reference faults in go2goal otherwise, rrt.py:317-318 / docs/getting-started.rst:36)."""
    from scipy import ndimage

    lab, k = ndimage.label(og == 0, structure=np.ones((3, 3), dtype=int))
    if k == 0:
        raise ValueError("occupancy grid has no free cell")
    sizes = np.bincount(lab.ravel())
    sizes[0] = 0
    return lab == int(np.argmax(sizes))


def random_connected_pair(og: np.ndarray, rnd_gen: np.random.Generator):
    """Two free cells drawn like ``random_point_og`` (rrt.py:27-44) but restricted to the
    largest free component."""
    cells = np.argwhere(largest_free_component(og))
    a = cells[rnd_gen.integers(low=0, high=cells.shape[0])]
    b = cells[rnd_gen.integers(low=0, high=cells.shape[0])]
    return a, b


def random_connected_pairs(og: np.ndarray, rnd_gen: np.random.Generator, count: int):
    """`count` consecutive ``random_connected_pair`` draws (same generator consumption, same pairs) with the component
    labelling done once instead of once per pair."""
    cells = np.argwhere(largest_free_component(og))
    out = []
    for _ in range(count):
        a = cells[rnd_gen.integers(low=0, high=cells.shape[0])]
        b = cells[rnd_gen.integers(low=0, high=cells.shape[0])]
        out.append((a, b))
    return out


class DeviceGrids:
    """`frames` seeded noise occupancy grids generated ON THE DEVICE and kept resident in HBM
    (the device-resident counterpart of ``perlin_occupancygrid(w, h, thresh, frames)``, reference oggen.py:7-45).

    ``host`` is the (frames, w, h) int64 copy the planner needs for ``free = argwhere(og == 0)``; it is
    bit-identical to ``perlin_occupancygrid(w, h, thresh, frames, seed)`` (tests/test_gpu_parity.py).
    ``planner.set_og_resident(grids, k)`` switches the planner to frame k without any upload
    (the per-frame ``set_og`` of anim.py:92-93)."""

    def __init__(self, ctx, w: int, h: int, thresh: float = 0.33, frames: int = 1, seed: int = 1, base_cell: float = None,
                 octaves: int = 4):
        lat = noise_lattices(w, h, frames, seed, base_cell, octaves)
        dims = [g.shape[:3] for _, _, g in lat]
        grads = np.concatenate([g.reshape(-1) for _, _, g in lat])
        self.ctx = ctx
        self.host = ctx.noise_grids(w, h, frames, np.float32(thresh), dims, [c for c, _, _ in lat], [a for _, a, _ in lat], grads).astype(np.int64)
        self.frames = frames
        self.generation = ctx.grid_generation()  # the frames live in the context's grid storage: any later upload evicts them

    def valid(self) -> bool:
        """False once something else (set_grid via planner.set_og + plan, another DeviceGrids) rewrote the context's grid."""
        return self.ctx.grid_generation() == self.generation

    def select(self, k: int):
        if not self.valid():
            raise RuntimeError("the device frames of this DeviceGrids are gone: the context's grid was uploaded or regenerated "
                               "since (set_og + plan, or another DeviceGrids on the same context); generate them again")
        self.ctx.select_frame(k)
