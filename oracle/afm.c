/* Oracle (test infrastructure only): CPU restatement of the reference's attraction-field-map kernel,
 * pixelspointspolygons/models/hisup/afm_module/afm_op/cuda/afm.cu:29-84 (CUDA only: it cannot be built or run in this image, so parity
 * is "unpinned": this file follows the source expression by expression - float variables, the double literal 1e-6 promoting the
 * projection division and the log encoding to double, strict `<` so the first of equally close segments wins - with the fused
 * multiply-adds nvcc's default contraction (--fmad=true) forms written out as fmaf). */
#include <math.h>
#include <stdint.h>

static float sgn(float x) { return x > 0 ? 1.0f : -1.0f; }

int p3o_afm(const float* lines, const int32_t* shape_info, int num, int height, int width, float* afmap, int32_t* aflabel) {
    for (int64_t index = 0; index < (int64_t)num * height * width; ++index) {
        int w = (int)(index % width), h = (int)((index / width) % height), n = (int)(index / width / height);
        int64_t x_index = (int64_t)n * 2 * height * width + (int64_t)h * width + w;
        int64_t y_index = x_index + (int64_t)height * width;
        int64_t label_index = (int64_t)n * height * width + (int64_t)h * width + w;
        float px = (float)w, py = (float)h;
        int start = shape_info[n * 4], end = shape_info[n * 4 + 1];
        float min_dis = 1e30f;
        afmap[x_index] = 0.f; afmap[y_index] = 0.f; aflabel[label_index] = 0;      /* at::zeros */
        for (int i = start; i < end; ++i) {
            float xs = (float)width / (float)shape_info[n * 4 + 3];
            float ys = (float)height / (float)shape_info[n * 4 + 2];
            float x1 = lines[4 * i] * xs, y1 = lines[4 * i + 1] * ys, x2 = lines[4 * i + 2] * xs, y2 = lines[4 * i + 3] * ys;
            float dx = x2 - x1, dy = y2 - y1;
            float norm2 = fmaf(dx, dx, dy * dy);
            float t = (float)((double)fmaf(px - x1, dx, (py - y1) * dy) / ((double)norm2 + 1e-6));
            t = t < 1.0 ? t : 1.0f;
            t = t > 0.0 ? t : 0.0f;
            float ax = fmaf(t, x2 - x1, x1) - px;
            float ay = fmaf(t, y2 - y1, y1) - py;
            float dis = fmaf(ax, ax, ay * ay);
            if (dis < min_dis) {
                min_dis = dis;
                afmap[x_index] = (float)(-(double)sgn(ax) * log((double)fabsf(ax / (float)width) + 1e-6));
                afmap[y_index] = (float)(-(double)sgn(ay) * log((double)fabsf(ay / (float)height) + 1e-6));
                aflabel[label_index] = i - start;
            }
        }
    }
    return 0;
}
