"""Installs the drop-in modules (`pointnet2_batch_cuda`, `chamfer`) and the `adaptpoint_amd`
package after building libadaptpoint_amd.so in-tree with hipcc for gfx950."""
from setuptools import find_packages, setup
from setuptools.command.build_py import build_py


class BuildWithHip(build_py):
    def run(self):
        from adaptpoint_amd import build as apn_build
        apn_build.build()
        super().run()


setup(
    name="adaptpoint_amd",
    version="0.1.0",
    description="MI355X-native set-abstraction hot path for AdaptPoint / OpenPoints",
    packages=find_packages(include=["adaptpoint_amd", "adaptpoint_amd.*"]),
    py_modules=["pointnet2_batch_cuda", "chamfer"],
    package_data={"adaptpoint_amd": ["libadaptpoint_amd.so", "csrc/*.hip", "csrc/*.h"]},
    cmdclass={"build_py": BuildWithHip},
)
