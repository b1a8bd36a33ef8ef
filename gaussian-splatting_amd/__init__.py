"""gaussian-splatting_amd: MI355X (gfx950) native render path behind the API of
ChristianLin0420/gaussian-splatting's GaussianSplat (importable as `gaussian_splatting_amd`)."""
from .model import GaussianSplat
from .render import render_views, RenderFunction

__all__ = ["GaussianSplat", "render_views", "RenderFunction"]
