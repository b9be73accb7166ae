"""surreal_amd -- the MI355X (gfx950) hot path of SurrealAI/surreal behind its Agent / Replay / Learner
plugin API.  The arithmetic lives in csrc/*.hip behind the C ABI of include/surreal_amd.h
(kernels.HipKernels is the only facade over it); everything else is the Python host side."""
