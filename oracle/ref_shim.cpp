// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
//
// Thin extern "C" wrapper that compiles the *unmodified* reference rasterizer
// (/root/reference/C++/DifferentiableRenderer.h, included from where it lies; nothing is
// copied into this repository) into oracle/_ref/libdeodr_ref.so.  It replaces the Cython
// marshalling layer (reference deodr/differentiable_renderer_cython.pyx:50-202, 206-410)
// with a flat POD struct so the oracle can be driven through ctypes, and converts the
// reference's `throw "literal"` (H.h:810, 2667-2712, 2924) into an error code because the
// Cython shim has no `except +` and would std::terminate.
//
// Built twice by oracle/Makefile:
//   libdeodr_ref.so          header as shipped
//   libdeodr_ref_fixed.so    same, with -DDEODR_REF_HEADER pointing at a temp copy with two adjoint
//                            defects repaired (oracle/Makefile: D1 texture_b overwrite H.h:621-624,
//                            D2 missing A0y_B fold in rasterize_edge_interpolated_error_B H.h:2595).
// The header uses SHRT_MAX without including <climits>; the Cython build gets it through
// Python.h -> <limits.h>.  Same macro, same value.
#include <climits>
#ifndef DEODR_REF_HEADER
#define DEODR_REF_HEADER "DifferentiableRenderer.h"
#endif
#include DEODR_REF_HEADER

extern "C" {

struct RefSceneFlat {
	const unsigned int *faces;
	const unsigned int *faces_uv;
	const double *depths;
	const double *uv;
	const double *ij;
	const double *shade;
	const double *colors;
	const unsigned char *edgeflags;
	const unsigned char *textured;
	const unsigned char *shaded;
	const double *texture;
	const double *background_image; // NULL xor background_color
	const double *background_color;
	double *uv_b;
	double *ij_b;
	double *shade_b;
	double *colors_b;
	double *texture_b;
	int nb_triangles;
	int nb_vertices;
	int nb_uv;
	int height;
	int width;
	int nb_colors;
	int texture_height;
	int texture_width;
	int clockwise;
	int backface_culling;
	int strict_edge;
	int perspective_correct;
	int integer_pixel_centers;
};

static const char *g_last_error = "";

static Scene to_scene(const RefSceneFlat *f)
{
	Scene s;
	s.faces = const_cast<unsigned int *>(f->faces);
	s.faces_uv = const_cast<unsigned int *>(f->faces_uv);
	s.depths = const_cast<double *>(f->depths);
	s.uv = const_cast<double *>(f->uv);
	s.ij = const_cast<double *>(f->ij);
	s.shade = const_cast<double *>(f->shade);
	s.colors = const_cast<double *>(f->colors);
	// the reference reinterprets uint8 arrays as bool* (pyx:150-152)
	s.edgeflags = reinterpret_cast<bool *>(const_cast<unsigned char *>(f->edgeflags));
	s.textured = reinterpret_cast<bool *>(const_cast<unsigned char *>(f->textured));
	s.shaded = reinterpret_cast<bool *>(const_cast<unsigned char *>(f->shaded));
	s.nb_triangles = f->nb_triangles;
	s.nb_vertices = f->nb_vertices;
	s.clockwise = f->clockwise != 0;
	s.backface_culling = f->backface_culling != 0;
	s.nb_uv = f->nb_uv;
	s.height = f->height;
	s.width = f->width;
	s.nb_colors = f->nb_colors;
	s.texture = const_cast<double *>(f->texture);
	s.texture_height = f->texture_height;
	s.texture_width = f->texture_width;
	s.background_image = const_cast<double *>(f->background_image);
	s.background_color = const_cast<double *>(f->background_color);
	s.uv_b = f->uv_b;
	s.ij_b = f->ij_b;
	s.shade_b = f->shade_b;
	s.colors_b = f->colors_b;
	s.texture_b = f->texture_b;
	s.strict_edge = f->strict_edge != 0;
	s.perspective_correct = f->perspective_correct != 0;
	s.integer_pixel_centers = f->integer_pixel_centers != 0;
	return s;
}

const char *deodr_ref_last_error(void) { return g_last_error; }

int deodr_ref_render_scene(const RefSceneFlat *f, double *image, double *z_buffer, double sigma,
						   int antialiase_error, double *obs, double *err_buffer)
{
	try
	{
		renderScene(to_scene(f), image, z_buffer, sigma, antialiase_error != 0, obs, err_buffer);
	}
	catch (const char *msg)
	{
		g_last_error = msg;
		return 1;
	}
	return 0;
}

int deodr_ref_render_scene_b(const RefSceneFlat *f, double *image, double *z_buffer, double *image_b,
							 double sigma, int antialiase_error, double *obs, double *err_buffer,
							 double *err_buffer_b)
{
	try
	{
		renderScene_B(to_scene(f), image, z_buffer, image_b, sigma, antialiase_error != 0, obs,
					  err_buffer, err_buffer_b);
	}
	catch (const char *msg)
	{
		g_last_error = msg;
		return 1;
	}
	return 0;
}

} // extern "C"
