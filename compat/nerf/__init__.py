"""Stand-in for the reference's `nerf` package ("importable as nerf.*").

Put this directory's PARENT (`compat/`) on `sys.path` / `PYTHONPATH` next to the repository root and the reference's entry scripts
(`train.py:12-20`, `ddp_train.py:17-25`, `model_average.py:16-27`) import the HIP path unchanged:

    PYTHONPATH=/path/to/repo:/path/to/repo/compat python train.py ...

Every module here is a two-line re-export of the `nerf_amd` module of the same name -- no logic lives in this package.  The list of
names the entry scripts need is golden G23 (tests/golden/g23_entry_imports.json, written from the reference by make_golden.py);
tests/test_abi_and_host.py checks the list on CPU, tests/test_gpu_parity.py imports and renders through it on the GPU."""
import nerf_amd as _nerf_amd        # fails loudly when libnerf_amd.so is not built: there is no CPU fallback behind this package

__all__ = ["procedures", "nerf_base", "mip_model", "addtional", "utils", "mip_methods", "nerf_helper", "ref_model", "ref_func", "param_com",
           "local_shuffler", "dataset", "timer"]
