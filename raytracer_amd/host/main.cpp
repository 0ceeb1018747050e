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
// --inflight K  K frames of an animation in flight at once (default 2): K pipelines, each with its own
//     loaded scene, device copy and host thread, take the frames in turn.  The last ray
//     generations of a frame are a few long rays in an otherwise idle GPU; the next frame's full
//     generations run beside them (camera_around_david stand-in on one MI355X: 2.4 -> 1.5 ms per frame
//     with three in flight).
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
    int tile_w = 64, tile_h = 8, res_w = 0, res_h = 0, inflight = 2;
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
    std::atomic<int> rc_all{0};
    Writer writer;
    // this process's frames, dealt to the pipelines in turn
    std::vector<size_t> mine;
    for (size_t si = 0; si < scenes.size(); si++)
        if ((int)(si % (size_t)shard_n) == shard_r) mine.push_back(si);
    const int n_pipes = (int)std::max<size_t>(1, std::min<size_t>((size_t)inflight, mine.size()));
    auto pipeline = [&](const int pipe) {
    rtx_host_scene* hs = nullptr;
    std::vector<rtx_scene*> ds;                 // the scene on every device taking part
    auto drop_device_scenes = [&ds] { for (rtx_scene* d : ds) rtx_scene_destroy(d); ds.clear(); };
    double upload_ms = 0;
    for (size_t mi = (size_t)pipe; mi < mine.size(); mi += (size_t)n_pipes) {
        const size_t si = mine[mi];
        int reused = 0;
        if (hs) {
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
            for (int k = 0; k < n_dev; k++) {
                rtx_scene* d = nullptr;
                if (rtx_scene_create(desc, devices[(size_t)k], &d) != RTX_OK) {
                    fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                    drop_device_scenes();
                    rtx_host_free(hs);
                    rc_all = 1;
                    return;                                  // no device: nothing else can work either
                }
                if (consistent) rtx_scene_set_consistent(d, 1);
                rtx_scene_set_generation_timing(d, 0);       // (the results file does not report per-generation times)
                ds.push_back(d);
            }
            upload_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        double total_ms = pre_ms;
        for (int k = 0; k < rtx_host_camera_count(hs); k++) {
            rtx_camera cam;
            char name[512];
            rtx_host_camera(hs, k, &cam, name, sizeof name);
            if (res_w > 0) { cam.width = res_w; cam.height = res_h; }
            OutputJob job;
            job.img.resize((size_t)cam.width * cam.height * 3);
            rtx_stats st;
            if (n_dev == 1) {
                if (rtx_render(ds[0], &cam, cam.width, cam.height, 0, 1, job.img.data(), 0, nullptr, &st) != RTX_OK) {
                    fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                    rc_all = 1;
                    break;
                }
            } else {
                // one host thread per device: its interleaved tiles into a packed buffer of its own
                auto t0 = std::chrono::high_resolution_clock::now();
                std::vector<std::vector<uint8_t>> packed((size_t)n_dev);
                std::vector<rtx_stats> part((size_t)n_dev);
                std::vector<std::string> errors((size_t)n_dev);
                std::vector<std::thread> workers;
                for (int r = 0; r < n_dev; r++) {
                    const int64_t bytes = rtx_render_output_bytes(&cam, tile_w, tile_h, r, n_dev);
                    packed[(size_t)r].resize((size_t)std::max<int64_t>(bytes, 0));
                    workers.emplace_back([&, r] {
                        memset(&part[(size_t)r], 0, sizeof(rtx_stats));
                        if (packed[(size_t)r].empty()) return;
                        if (rtx_render(ds[(size_t)r], &cam, tile_w, tile_h, r, n_dev, packed[(size_t)r].data(), 0, nullptr,
                                       &part[(size_t)r]) != RTX_OK)
                            errors[(size_t)r] = rtx_last_error();      // (the error text is per thread)
                    });
                }
                for (std::thread& t : workers) t.join();
                bool ok = true;
                memset(&st, 0, sizeof st);
                for (int r = 0; r < n_dev; r++) {
                    if (!errors[(size_t)r].empty()) { fprintf(stderr, "%s: device %d: %s\n", scenes[si].c_str(), devices[(size_t)r], errors[(size_t)r].c_str()); ok = false; continue; }
                    if (!packed[(size_t)r].empty() &&
                        rtx_unpack_tiles(packed[(size_t)r].data(), cam.width, cam.height, tile_w, tile_h, r, n_dev, job.img.data()) != RTX_OK) {
                        fprintf(stderr, "%s: %s\n", scenes[si].c_str(), rtx_last_error());
                        ok = false;
                    }
                    st.rays += part[(size_t)r].rays; st.primary_rays += part[(size_t)r].primary_rays;
                    st.shadow_rays += part[(size_t)r].shadow_rays; st.secondary_rays += part[(size_t)r].secondary_rays;
                    st.kernel_ms = std::max(st.kernel_ms, part[(size_t)r].kernel_ms);
                }
                if (!ok) { rc_all = 1; break; }
                st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
            }
            total_ms += st.total_ms;
            job.base = out_dir + name;
            job.stem = job.base.substr(0, job.base.find_last_of('.'));
            job.width = cam.width; job.height = cam.height; job.png = png;
            char buf[1536];
            snprintf(buf, sizeof buf,
                     "{\n    \"enableBackFaceCulling\": %s,\n    \"isMultiThreaded\": %s,\n"
                     "    \"preprocessingTimeMs\": %.3f,\n    \"renderTimeMs\": %.3f,\n    \"sceneName\": \"%s\",\n"
                     "    \"totalTimeMs\": %.3f,\n    \"useBVH\": %s,\n    \"device\": \"hip:%d\",\n"
                     "    \"kernelTimeMs\": %.3f,\n    \"parseTimeMs\": %.3f,\n    \"uploadTimeMs\": %.3f,\n"
                     "    \"geometryReused\": %s,\n    \"nGpus\": %d,\n    \"rays\": %llu,\n    \"MraysPerS\": %.3f,\n"
                     "    \"width\": %d,\n    \"height\": %d\n}",
                     cull ? "true" : "false", multithreaded ? "true" : "false", pre_ms, st.total_ms, name, total_ms,
                     use_bvh ? "true" : "false", device, st.kernel_ms, parse_ms, upload_ms, reused ? "true" : "false",
                     n_dev, (unsigned long long)st.rays, st.total_ms > 0 ? (double)st.rays / st.total_ms / 1e3 : 0.0,
                     cam.width, cam.height);
            job.results = buf;
            writer.push(std::move(job));
            if (!quiet)
                printf("%s: %dx%d rendered in %.3f ms (kernels %.3f ms), parse %.1f ms, preprocess %.1f ms, upload %.1f ms%s\n",
                       name, cam.width, cam.height, st.total_ms, st.kernel_ms, parse_ms, pre_ms, upload_ms,
                       reused ? " (geometry kept)" : "");
        }
    }
    drop_device_scenes();
    if (hs) rtx_host_free(hs);
    };
    std::vector<std::thread> pipes;
    for (int p = 1; p < n_pipes; p++) pipes.emplace_back(pipeline, p);
    pipeline(0);
    for (std::thread& t : pipes) t.join();
    if (!writer.finish()) rc_all = 1;
    return rc_all;
}
