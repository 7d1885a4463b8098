"""take_amd — host side of the MI355X path-tracing core behind TaKe's `render()`.

  capi    ctypes binding of the C ABI (include/take_hip.h, take_amd/libtake_hip.so built from csrc/)
  cdefs   ctypes mirror of the header's structs and constants
  scene   SceneData: a flattened reference `Scene` (+ .tkscene files written by take_amd/host/take_flatten.hpp)
  render  the reference's render()/main() convention on top of the C ABI;  exr: its image.exr writer
  dist    row-strip sharding over one process per GPU and the single gather
  scenes  procedural benchmark scenes (BASELINE.json configs)
There is no CPU rendering path in this package: every render call goes to the HIP library and fails without a GPU.
"""
