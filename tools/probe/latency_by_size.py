import sys, time, statistics
sys.path.insert(0, '.')
import torch, jpeggpu_amd as jp
from tools import jpegsynth
S420 = ((2, 2), (1, 1), (1, 1))
inputs = [((w, h), jpegsynth.encode(w, h, S420, restart_interval=(w + 15) // 16, quality=88, noise=9, seed=5))
          for (w, h) in ((320, 240), (640, 480), (1280, 720), (1920, 1080), (4032, 3024))]
inputs.append((("photo", "IMG_6510"), open('tests/golden/IMG_6510.JPG', 'rb').read()))
inputs.append((("cfg5", "no dri"), jpegsynth.config(5)))
for (w, h), data in inputs:
    row = []
    for sb in (32, 64, 128, 256):
        dec = jp.Decoder(sb)
        pinned = torch.empty(len(data), dtype=torch.uint8).pin_memory(); pinned.numpy()[:] = memoryview(data)
        info = dec.parse_header(pinned.data_ptr(), pinned.numel()); n = dec.get_buffer_size()
        tmp = torch.empty(n + 256, dtype=torch.uint8, device='cuda'); base = (tmp.data_ptr() + 255) // 256 * 256
        planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device='cuda') for c in range(info.num_components)]
        ptrs = [p.data_ptr() for p in planes]; pit = [p.stride(0) for p in planes]
        st = torch.cuda.Stream()
        lat = []
        for it in range(60):
            t = time.perf_counter()
            dec.parse_header(pinned.data_ptr(), pinned.numel()); n = dec.get_buffer_size()
            dec.transfer(base, n, st.cuda_stream); dec.decode(ptrs, pit, base, n, st.cuda_stream); st.synchronize()
            if it >= 10: lat.append((time.perf_counter() - t) * 1e3)
        row.append(round(statistics.median(lat), 3))
        dec.cleanup()
    print(w, h, len(data), "p50 ms for sb 32/64/128/256:", row)
