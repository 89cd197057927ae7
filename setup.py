"""Install wayne_amd with the reference's command name:  wayne -p params.yml

    python setup.py build_hip      # hipcc --offload-arch=gfx950 -> wayne_amd/libwayne_hip.so (in tree)
    pip install -e .               # `wayne` on PATH

The HIP library is built in tree by `python -m wayne_amd.build` (no CPU fallback exists; the package
imports without a GPU but every compute entry point needs an MI355X).
"""
from setuptools import Command, find_packages, setup


class BuildHip(Command):
    description = "compile the gfx950 kernels and the C ABI into wayne_amd/libwayne_hip.so"
    user_options = []

    def initialize_options(self):
        pass

    def finalize_options(self):
        pass

    def run(self):
        from wayne_amd import build
        build.build()


setup(
    name="wayne_amd",
    version="0.1.0",
    description="MI355X-native WFC3-IR exposure synthesis (the hot path of ucl-exoplanets/wayne)",
    packages=find_packages(include=["wayne_amd", "wayne_amd.*"]),
    package_data={"wayne_amd": ["libwayne_hip.so", "data/*", "csrc/*"]},
    install_requires=["numpy", "pyyaml"],
    entry_points={"console_scripts": ["wayne=wayne_amd.run_visit:run"]},
    cmdclass={"build_hip": BuildHip},
)
