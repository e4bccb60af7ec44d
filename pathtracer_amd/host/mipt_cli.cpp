// mipt_render — a GUI-free front end on top of the host mirror, in the spirit of the reference's command line
// (`pathtracer scene.scn out.png`, mainApp.cpp:38-49): the default loadScene() scene (light, environment sphere, ground
// plane, camera) plus one OBJ/MTL mesh placed like a file dropped on the GUI (scale 30, bottom on the plane,
// mainApp.cpp:2402-2410), rendered with Raytracer::render_image_nopreviz() on the GPU and written by save_image's rule:
// the container follows the extension of the output name (utils.cpp:178-234; .png / .bmp / .tga / .jpg / .ppm / .hdr are written,
// any other extension is an error, never another format under that name).
//
//   mipt_render scene.scn out.png [nameSubst] [-s WxH] [-n spp] [-b bounces] [-f frame] [-d device[,device...]] [-g gpus]
//        -f: Scene::current_frame, the time the scene's key-framed transforms are evaluated at (default 0)
//        -d 0,1,2,3 or -g 4: the devices rendering the frame (tiles dealt to them, one RCCL reduce; mipt_create with n > 1)
//        the reference's command line (mainApp.cpp:38-49): loadScene(), load_scene(argv[1][, argv[3]]), render_image_nopreviz(),
//        save_image(argv[2]); nameSubst replaces the '#' in the mesh file names of the scene (Geometry.h:524-526); options override the file
//   mipt_render mesh.obj out.png [-s WxH] [-n spp] [-b bounces] [-d device] [--merl file.binary] [--mirror]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "mipt_host.h"

using namespace mipt_host;

int main(int argc, char** argv) {
	if (argc < 3) { fprintf(stderr, "usage: %s scene.scn|mesh.obj out.png|.bmp|.tga|.jpg|.ppm|.hdr [nameSubst] [-s WxH] [-n spp] [-b bounces] [-f frame] [-d device,...] [-g gpus] [--merl file.binary] [--mirror]\n", argv[0]); return 2; }
	int W = 1000, H = 800, spp = 100, bounces = 3, frame = 0;
	int devices[64] = {0}, ndev = 1;
	const char* merl = nullptr;
	bool mirror = false;
	const char* name_subst = nullptr;                    // argv[3] of the reference's command line, when it is not an option
	int first_opt = 3;
	if (argc > 3 && argv[3][0] != '-') { name_subst = argv[3]; first_opt = 4; }
	// an output name no writer exists for is refused before anything is rendered (by its name: an existing file of that
	// name is not touched until the final write)
	if (!mh_image_format_supported(argv[2], 1)) { fprintf(stderr, "%s: no writer for this extension (.png, .bmp, .tga, .jpg, .ppm, .hdr)\n", argv[2]); return 2; }
	for (int i = first_opt; i < argc; i++) {
		if (!strcmp(argv[i], "-s") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &W, &H) != 2) { fprintf(stderr, "bad size\n"); return 2; } }
		else if (!strcmp(argv[i], "-n") && i + 1 < argc) spp = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-b") && i + 1 < argc) bounces = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-f") && i + 1 < argc) frame = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-d") && i + 1 < argc) {          // one device or a comma-separated list
			ndev = 0;
			for (const char* q = argv[++i]; *q && ndev < 64;) { devices[ndev++] = atoi(q); while (*q && *q != ',') q++; if (*q == ',') q++; }
			if (ndev == 0) { fprintf(stderr, "bad device list\n"); return 2; }
		}
		else if (!strcmp(argv[i], "-g") && i + 1 < argc) { ndev = atoi(argv[++i]); if (ndev < 1 || ndev > 64) { fprintf(stderr, "bad GPU count\n"); return 2; } for (int k = 0; k < ndev; k++) devices[k] = k; }
		else if (!strcmp(argv[i], "--merl") && i + 1 < argc) merl = argv[++i];
		else if (!strcmp(argv[i], "--mirror")) mirror = true;
		else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
	}
	mh_raytracer* h = mh_create();                                       // new Raytracer + loadScene()
	int rc = mh_open_devices(h, devices, ndev);
	if (rc != MIPT_OK) { fprintf(stderr, "cannot open GPU %d%s (status %d): %s\n", devices[0], ndev > 1 ? ", ..." : "", rc, mh_last_error(h)); return 1; }   // no CPU fallback
	auto t0 = std::chrono::steady_clock::now();
	const size_t len = strlen(argv[1]);
	if (len > 4 && !strcmp(argv[1] + len - 4, ".scn")) {                   // Raytracer::load_scene; explicit options override the file
		if (mh_load_scene_subst(h, argv[1], name_subst) != 0) { fprintf(stderr, "%s\n", mh_last_error(h)); return 1; }
		float hdr[32]; mh_get_scene_header(h, hdr);
		bool sized = false, sampled = false, bounced = false;
		for (int i = first_opt; i < argc; i++) { sized |= !strcmp(argv[i], "-s"); sampled |= !strcmp(argv[i], "-n"); bounced |= !strcmp(argv[i], "-b"); }
		if (!sized) { W = (int)hdr[0]; H = (int)hdr[1]; }
		if (!sampled) spp = (int)hdr[2];
		if (!bounced) bounces = (int)hdr[3];
		mh_set_render(h, W, H, spp, bounces, hdr[16]);
	} else {
		mh_set_render(h, W, H, spp, bounces, 0.5f);
		int obj = mh_add_mesh_obj(h, argv[1], 30.f, 1);
		if (obj < 0) { fprintf(stderr, "%s\n", mh_last_error(h)); return 1; }
		if (mirror) mh_set_object_flags(h, obj, 1, 0);
		if (merl && mh_set_brdf_merl_file(h, obj, merl) != 0) { fprintf(stderr, "%s\n", mh_last_error(h)); return 1; }
	}
	mh_set_frame(h, frame);
	auto t1 = std::chrono::steady_clock::now();
	rc = mh_render_image_nopreviz(h);
	auto t2 = std::chrono::steady_clock::now();
	if (rc != MIPT_OK) { fprintf(stderr, "render failed (status %d): %s\n", rc, mh_last_error(h)); return 1; }
	{
		char why[256];
		int bad;
		if (mh_image_format_supported(argv[2], 0)) bad = mh_save_image(argv[2], mh_image(h), W, H, why, sizeof why);   // save_image(argv[2], &raytracer.image[0], W, H)
		else {
			// .hdr: the reference hands its 8-bit buffer to the float encoder (utils.cpp:184-190 instantiated for unsigned char reads
			// W*H*3 floats out of W*H*3 bytes); here the file holds the linear radiance image, imagedouble / sample_count, white = 1
			std::vector<float> lin((size_t)W * H * 3);
			const float* acc = mh_imagedouble(h); const float* cnt = mh_sample_count(h);
			for (size_t i = 0; i < (size_t)W * H; i++) for (int k = 0; k < 3; k++) lin[3 * i + k] = cnt[i] > 0.f ? acc[3 * i + k] / cnt[i] / 196964.7f : 0.f;
			bad = mh_save_image_f32(argv[2], lin.data(), W, H, 1.f, why, sizeof why);
		}
		if (bad) { fprintf(stderr, "%s: %s\n", argv[2], why); return 1; }
	}
	auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
	fprintf(stderr, "%s: load + BVH %.2f s, render %dx%d x %d spp %.2f s -> %s\n", argv[1], secs(t0, t1), W, H, spp, secs(t1, t2), argv[2]);
	mh_destroy(h);
	return 0;
}
