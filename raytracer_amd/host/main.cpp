// raytracer_amd -- command-line host with the reference's flags (raytracer.cpp:173-210)
// that renders on the MI355X core instead of the CPU loops.
//
//   raytracer_amd <scene.json> [more.json ...] [-b [0|1]] [-m [0|1]] [-c [0|1]]
//                 [--device D | --gpus N | --devices a,b,..] [--tile WxH] [--resolution WxH]
//                 [--shard R/N] [--inflight K] [--out DIR] [--png] [--quiet] [--gpu-build] [--consistent]
//
// -b  BVH on/off (default on; `-b` alone = on).  Off ships meshes without a BVH; the device
//     then tests every face of a mesh in face order, like the reference's brute-force loop.
// -m  accepted for compatibility (the reference's thread switch); it has no effect here.
// -c  back-face culling (default off; `-c` alone = off, `-c 1` = on), as in the reference.
// --gpu-build  build the meshes' BVHs on the device (rtx_build_bvh) instead of on the host.
// --consistent  the core's consistent mode (rtx_scene_set_consistent): scaled instances are searched
//     with the bound in object space, shadow rays end at their first hit.  Not the reference's
//     pictures where a scene has scaled instances.
// --shard R/N  render only scene files i with i % N == R (whole-frame sharding of an
//     animation across N processes, one per GPU; no communication needed).
// --gpus N / --devices a,b,..  split every frame over several GPUs of this process: the scene is
//     uploaded to each, device k renders the interleaved tiles k, k + n, ... (--tile WxH, default
//     64x8) on a host thread of its own, and the packed tiles are assembled into the frame here
//     (the reference splits a frame into row bands over its threads, raytracer.cpp:107-126;
//     interleaved tiles balance scenes whose geometry sits in one strip).  A device may be named
//     more than once (two tile streams on one GPU).
// --inflight K  up to K pictures in flight at once (default 3) on ONE loaded scene and ONE device copy
//     of it: K frame contexts (rtx_frame_create) of the device scene take the cameras and animation
//     frames in turn, each submitted without waiting (rtx_render_async) and collected when its turn
//     comes round again.  The last ray generations of a frame are a few long rays in an otherwise
//     idle GPU; the next frame's full generations run beside them.  Geometry is uploaded once; a
//     frame keeps the lights that were current when it was submitted.
// --resolution WxH  render every camera at this resolution instead of the scene file's
//     ImageResolution (same view: keep the aspect ratio).
//
// For every camera it writes <out>/<ImageName stem>.ppm (ASCII P3, byte-identical layout to
// scene.cpp:675-697), optionally a .png, and <stem>_results.json with the reference's keys
// (raytracer.cpp:269-282) plus device timings.
//
// Several scene files form an animation (the reference renders them one process per frame,
// runner.py:105-141).  Here they go through one pipeline: a frame whose geometry equals its
// predecessor's keeps the loaded scene and its device copy and only takes over cameras and
// lights (rtx_host_load_next + rtx_scene_update_lights), and pictures are encoded and
// written by a second thread while the next frame renders.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rtx_amd.h"
#include "rtx_host.h"

namespace {

// One finished picture on its way to disk.
struct OutputJob {
    std::string base, stem, results;
    std::vector<uint8_t> img;
    int width = 0, height = 0;
    bool png = false;
};

// Writes pictures in the background, kThreads files at a time (an ASCII P3 file takes longer to
// format than its frame takes to render); at most kDepth pictures wait (each is a full frame).
class Writer {
public:
    Writer() { for (int i = 0; i < kThreads; i++) workers_.emplace_back([this] { run(); }); }
    ~Writer() { finish(); }
    void push(OutputJob&& job) {
        std::unique_lock<std::mutex> lk(m_);
        room_.wait(lk, [this] { return queue_.size() < kDepth; });
        queue_.push_back(std::move(job));
        work_.notify_one();
    }
    bool finish() {                       // true when every file was written
        {
            std::lock_guard<std::mutex> lk(m_);
            if (done_) return !failed_;
            done_ = true;
        }
        work_.notify_all();
        for (std::thread& t : workers_) t.join();
        return !failed_;
    }

private:
    static constexpr size_t kDepth = 8;
    static constexpr int kThreads = 4;
    void run() {
        while (true) {
            OutputJob job;
            {
                std::unique_lock<std::mutex> lk(m_);
                work_.wait(lk, [this] { return !queue_.empty() || done_; });
                if (queue_.empty()) return;
                job = std::move(queue_.front());
                queue_.pop_front();
                room_.notify_one();
            }
            if (rtx_host_write_ppm(job.base.c_str(), job.img.data(), job.width, job.height) != 0) {
                fprintf(stderr, "%s\n", rtx_host_last_error());
                failed_ = true;
            }
            if (job.png && rtx_host_write_png((job.stem + ".png").c_str(), job.img.data(), job.width, job.height) != 0) {
                fprintf(stderr, "%s\n", rtx_host_last_error());
                failed_ = true;
            }
            if (FILE* f = fopen((job.stem + "_results.json").c_str(), "w")) {
                fputs(job.results.c_str(), f);
                fclose(f);
            }
        }
    }
    std::mutex m_;
    std::condition_variable work_, room_;
    std::deque<OutputJob> queue_;
    bool done_ = false;
    std::atomic<bool> failed_{false};
    std::vector<std::thread> workers_;
};

}  // namespace

int main(int argc, char** argv) {
    std::vector<std::string> scenes;
    bool use_bvh = true, cull = false, multithreaded = true, png = false, quiet = false, gpu_build = false, consistent = false;
    int device = 0, shard_r = 0, shard_n = 1;
    std::vector<int> devices;
    int tile_w = 64, tile_h = 8, res_w = 0, res_h = 0, inflight = 3;
    std::string out_dir = "../my_outputs_hw2/";            // raytracer.cpp:13
    auto opt01 = [&](int& i, bool bare_value, bool& target) {
        if (i + 1 < argc && (!strcmp(argv[i + 1], "0") || !strcmp(argv[i + 1], "1"))) { target = atoi(argv[++i]) != 0; }
        else target = bare_value;
    };
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-m")) opt01(i, false, multithreaded);
        else if (!strcmp(argv[i], "-b")) opt01(i, true, use_bvh);
        else if (!strcmp(argv[i], "-c") || !strcmp(argv[i], "--no-cull")) opt01(i, false, cull);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) { out_dir = argv[++i]; if (!out_dir.empty() && out_dir.back() != '/') out_dir += '/'; }
        else if (!strcmp(argv[i], "--shard") && i + 1 < argc) { if (sscanf(argv[++i], "%d/%d", &shard_r, &shard_n) != 2 || shard_n < 1 || shard_r < 0 || shard_r >= shard_n) { fprintf(stderr, "bad --shard\n"); return 2; } }
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { devices.clear(); for (int k = 0, n = atoi(argv[++i]); k < n; k++) devices.push_back(k); }
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            devices.clear();
            for (const char* p = argv[++i]; *p;) { devices.push_back(atoi(p)); p += strcspn(p, ","); if (*p) p++; }
        }
        else if (!strcmp(argv[i], "--tile") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &tile_w, &tile_h) != 2 || tile_w < 1 || tile_h < 1) { fprintf(stderr, "bad --tile\n"); return 2; } }
        else if (!strcmp(argv[i], "--resolution") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &res_w, &res_h) != 2 || res_w < 1 || res_h < 1) { fprintf(stderr, "bad --resolution\n"); return 2; } }
        else if (!strcmp(argv[i], "--inflight") && i + 1 < argc) inflight = std::max(1, std::min(8, atoi(argv[++i])));
        else if (!strcmp(argv[i], "--png")) png = true;
        else if (!strcmp(argv[i], "--gpu-build")) gpu_build = true;
        else if (!strcmp(argv[i], "--consistent")) consistent = true;
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else scenes.push_back(argv[i]);
    }
    if (scenes.empty()) {
        fprintf(stderr, "usage: raytracer_amd <scene.json>... [-b [0|1]] [-m [0|1]] [-c [0|1]] [--device D | --gpus N | --devices a,b,..] "
                        "[--tile WxH] [--resolution WxH] [--shard R/N] [--inflight K] [--out DIR] [--png] [--gpu-build] [--consistent]\n");
        return 2;
    }
    (void)multithreaded;
    if (devices.empty()) devices.push_back(device);
    device = devices[0];
    const int n_dev = (int)devices.size();
    int rc_all = 0;
    Writer writer;

    rtx_host_scene* hs = nullptr;
    std::vector<rtx_scene*> ds;                 // the scene on every device taking part
    // A picture in flight: one frame context per device (one device: the whole picture; several: that
    // device's interleaved tiles, packed), and what is needed to finish it when its turn comes again.
    struct Slot {
        std::vector<rtx_frame*> fr;
        std::vector<std::vector<uint8_t>> packed;
        OutputJob job;
        rtx_camera cam{};
        std::string scene_file, name;
        double parse_ms = 0, pre_ms = 0, upload_ms = 0;
        bool reused = false, busy = false;
        size_t file = 0;
        std::chrono::high_resolution_clock::time_point t0;
    };
    std::vector<Slot> slots((size_t)inflight);
    size_t next_slot = 0;
    std::vector<double> file_ms(scenes.size(), 0.0);   // (the reference's totalTimeMs accumulates over a file's cameras, raytracer.cpp:260)

    // Wait for a slot's picture and hand it to the writer.
    auto collect = [&](Slot& sl) {
        if (!sl.busy) return;
        sl.busy = false;
        rtx_stats st;
        memset(&st, 0, sizeof st);
        bool ok = true;
        for (int r = 0; r < n_dev; r++) {
            rtx_stats part;
            if (n_dev > 1 && sl.packed[(size_t)r].empty()) continue;       // (more devices than tiles)
            if (rtx_frame_wait(sl.fr[(size_t)r], &part) != RTX_OK) {
                fprintf(stderr, "%s: device %d: %s\n", sl.scene_file.c_str(), devices[(size_t)r], rtx_last_error());
                ok = false;
                continue;
            }
            if (n_dev > 1 && rtx_unpack_tiles(sl.packed[(size_t)r].data(), sl.cam.width, sl.cam.height, tile_w, tile_h, r, n_dev,
                                              sl.job.img.data()) != RTX_OK) {
                fprintf(stderr, "%s: %s\n", sl.scene_file.c_str(), rtx_last_error());
                ok = false;
            }
            st.rays += part.rays; st.primary_rays += part.primary_rays;
            st.shadow_rays += part.shadow_rays; st.secondary_rays += part.secondary_rays;
            st.kernel_ms = std::max(st.kernel_ms, part.kernel_ms);
        }
        if (!ok) { rc_all = 1; return; }
        // submit -> picture complete (with pictures in flight this spans the neighbours' work too)
        st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - sl.t0).count();
        file_ms[sl.file] += st.total_ms;
        OutputJob& job = sl.job;
        job.base = out_dir + sl.name;
        job.stem = job.base.substr(0, job.base.find_last_of('.'));
        job.width = sl.cam.width; job.height = sl.cam.height; job.png = png;
        char buf[1536];
        snprintf(buf, sizeof buf,
                 "{\n    \"enableBackFaceCulling\": %s,\n    \"isMultiThreaded\": %s,\n"
                 "    \"preprocessingTimeMs\": %.3f,\n    \"renderTimeMs\": %.3f,\n    \"sceneName\": \"%s\",\n"
                 "    \"totalTimeMs\": %.3f,\n    \"useBVH\": %s,\n    \"device\": \"hip:%d\",\n"
                 "    \"kernelTimeMs\": %.3f,\n    \"parseTimeMs\": %.3f,\n    \"uploadTimeMs\": %.3f,\n"
                 "    \"geometryReused\": %s,\n    \"nGpus\": %d,\n    \"framesInFlight\": %d,\n    \"rays\": %llu,\n    \"MraysPerS\": %.3f,\n"
                 "    \"width\": %d,\n    \"height\": %d\n}",
                 cull ? "true" : "false", multithreaded ? "true" : "false", sl.pre_ms, st.total_ms, sl.name.c_str(), sl.pre_ms + file_ms[sl.file],
                 use_bvh ? "true" : "false", device, st.kernel_ms, sl.parse_ms, sl.upload_ms, sl.reused ? "true" : "false",
                 n_dev, inflight, (unsigned long long)st.rays, st.total_ms > 0 ? (double)st.rays / st.total_ms / 1e3 : 0.0,
                 sl.cam.width, sl.cam.height);
        job.results = buf;
        if (!quiet)
            printf("%s: %dx%d rendered in %.3f ms (kernels %.3f ms), parse %.1f ms, preprocess %.1f ms, upload %.1f ms%s\n",
                   sl.name.c_str(), sl.cam.width, sl.cam.height, st.total_ms, st.kernel_ms, sl.parse_ms, sl.pre_ms, sl.upload_ms,
                   sl.reused ? " (geometry kept)" : "");
        writer.push(std::move(job));
        sl.job = OutputJob();
    };
    auto drain = [&] { for (size_t k = 0; k < slots.size(); k++) collect(slots[(next_slot + k) % slots.size()]); };
    auto drop_device_scenes = [&] {
        drain();
        for (Slot& sl : slots) { for (rtx_frame* f : sl.fr) rtx_frame_destroy(f); sl.fr.clear(); }
        for (rtx_scene* d : ds) rtx_scene_destroy(d);
        ds.clear();
    };

    double upload_ms = 0;
    for (size_t si = 0; si < scenes.size(); si++) {
        if ((int)(si % (size_t)shard_n) != shard_r) continue;
        int reused = 0;
        if (hs) {
            // (pictures in flight read nothing of the host scene: the device has its own copy, and a
            // frame holds its camera and its version of the lights)
            rtx_host_scene* next = nullptr;
            if (rtx_host_load_next(scenes[si].c_str(), hs, &next, &reused) != 0) {
                fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_host_last_error());
                rc_all = 1;
                continue;                                    // the previous scene stays loaded
            }
            hs = next;
            // geometry replaced: the device copies belong to the previous scene, whatever happens next
            if (!reused) drop_device_scenes();
        } else if (rtx_host_load(scenes[si].c_str(), use_bvh && !gpu_build, cull, &hs) != 0) {
            fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_host_last_error());
            rc_all = 1;
            hs = nullptr;
            continue;
        }
        double parse_ms = 0, pre_ms = 0;
        rtx_host_timings(hs, &parse_ms, &pre_ms);
        const rtx_scene_desc* desc = rtx_host_desc(hs);
        if (use_bvh && gpu_build && !reused) {
            // the reference's trees (MeshBVH::build), built on the device and handed to the host scene
            bool ok = true;
            for (int m = 0; m < desc->n_meshes && ok; m++) {
                const rtx_mesh& mesh = desc->meshes[m];
                if (mesh.n_faces <= 0) continue;
                std::vector<rtx_bvh_node> nodes(2 * (size_t)mesh.n_faces);
                std::vector<int32_t> order((size_t)mesh.n_faces);
                int32_t n_nodes = 0;
                double ms = 0;
                if (rtx_build_bvh(device, desc->vertices, desc->n_vertices, mesh.faces, mesh.n_faces, nodes.data(), &n_nodes,
                                  order.data(), &ms) != RTX_OK) {
                    fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                    ok = false;
                } else if (rtx_host_attach_bvh(hs, m, nodes.data(), n_nodes, order.data()) != 0) {
                    fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_host_last_error());
                    ok = false;
                }
                pre_ms += ms;
            }
            if (!ok) { rc_all = 1; continue; }
        }
        if (reused && !ds.empty()) {
            upload_ms = 0;
            bool ok = true;
            for (rtx_scene* d : ds) ok = ok && rtx_scene_update_lights(d, desc->lights, desc->n_lights) == RTX_OK;
            if (!ok) {
                fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                rc_all = 1;
                continue;
            }
        } else {
            drop_device_scenes();
            auto t0 = std::chrono::high_resolution_clock::now();
            bool ok = true;
            for (int k = 0; k < n_dev && ok; k++) {
                rtx_scene* d = nullptr;
                ok = rtx_scene_create(desc, devices[(size_t)k], &d) == RTX_OK;
                if (ok) {
                    if (consistent) rtx_scene_set_consistent(d, 1);
                    rtx_scene_set_generation_timing(d, 0);       // (the results file does not report per-generation times)
                    ds.push_back(d);
                }
            }
            for (Slot& sl : slots)
                for (int k = 0; k < n_dev && ok; k++) {
                    rtx_frame* f = nullptr;
                    ok = rtx_frame_create(ds[(size_t)k], &f) == RTX_OK;
                    if (ok) sl.fr.push_back(f);
                }
            if (!ok) {
                fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                drop_device_scenes();
                rtx_host_free(hs);
                writer.finish();
                return 1;                                    // no device: nothing else can work either
            }
            upload_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        for (int k = 0; k < rtx_host_camera_count(hs); k++) {
            Slot& sl = slots[next_slot];
            next_slot = (next_slot + 1) % slots.size();
            collect(sl);                                     // the picture this slot was busy with
            char name[512];
            rtx_host_camera(hs, k, &sl.cam, name, sizeof name);
            if (res_w > 0) { sl.cam.width = res_w; sl.cam.height = res_h; }
            sl.name = name; sl.scene_file = scenes[si]; sl.file = si;
            sl.parse_ms = parse_ms; sl.pre_ms = pre_ms; sl.upload_ms = upload_ms; sl.reused = reused != 0;
            sl.job.img.resize((size_t)sl.cam.width * sl.cam.height * 3);
            sl.t0 = std::chrono::high_resolution_clock::now();
            bool ok = true;
            if (n_dev == 1) {
                ok = rtx_render_async(sl.fr[0], &sl.cam, sl.cam.width, sl.cam.height, 0, 1, sl.job.img.data(), 0, nullptr) == RTX_OK;
            } else {
                // every device its interleaved tiles into a packed buffer of its own, all submitted before any is waited for
                sl.packed.resize((size_t)n_dev);
                for (int r = 0; r < n_dev && ok; r++) {
                    const int64_t bytes = rtx_render_output_bytes(&sl.cam, tile_w, tile_h, r, n_dev);
                    sl.packed[(size_t)r].resize((size_t)std::max<int64_t>(bytes, 0));
                    if (sl.packed[(size_t)r].empty()) continue;
                    ok = rtx_render_async(sl.fr[(size_t)r], &sl.cam, tile_w, tile_h, r, n_dev, sl.packed[(size_t)r].data(), 0, nullptr) == RTX_OK;
                }
            }
            if (!ok) {
                fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                rc_all = 1;
                // (frames of this picture that were submitted are waited for and dropped)
                for (int r = 0; r < n_dev; r++) { rtx_stats st; (void)rtx_frame_wait(sl.fr[(size_t)r], &st); }
                break;
            }
            sl.busy = true;
        }
    }
    drop_device_scenes();
    if (hs) rtx_host_free(hs);
    if (!writer.finish()) rc_all = 1;
    return rc_all;
}
