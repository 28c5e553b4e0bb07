/* include/deodr_hip.h -- C ABI of libdeodr_hip.so, the MI355X (gfx950) implementation of DEODR's rasterizer hot path.
 *
 * Drop-in boundary.  The two compute entry points replace, argument for argument, the two
 * functions the reference's FFI shim binds:
 *
 *   deodr_hip_render_scene     <->  void renderScene  (Scene, double* image, double* z_buffer, double sigma,
 *                                                      bool antialiaseError, double* obs, double* err_buffer)
 *                                   /root/reference/C++/DifferentiableRenderer.h:2717, bound at
 *                                   deodr/differentiable_renderer_cython.pyx:44 and called at pyx:202
 *   deodr_hip_render_scene_b   <->  void renderScene_B(Scene, double* image, double* z_buffer, double* image_b,
 *                                                      double sigma, bool antialiaseError, double* obs,
 *                                                      double* err_buffer, double* err_buffer_b)
 *                                   DifferentiableRenderer.h:2903, bound at pyx:45 and called at pyx:405
 *   DeodrHipScene              <->  struct Scene, DifferentiableRenderer.h:56-90 (same fields, same meaning; pointers are
 *                                   DEVICE pointers, `bool` flags are int, plus dtype tags and a view count)
 *
 * Differences from the reference interface, all forced by the device:
 *   - every array pointer is a device pointer (HBM); nothing is copied to or from the host by this library;
 *   - the caller passes a `stream` (hipStream_t as void*) and a device workspace (size from deodr_hip_workspace_bytes);
 *     calls are asynchronous on that stream and never allocate (a call may run part of its kernels on ONE stream the library owns per
 *     device -- the background fill of a forward-only call, the head walkers of a textured fit step of 8 views or more -- forked from and
 *     joined back to `stream` by events inside the call: to the caller the call is ordered on `stream` alone; under stream capture the
 *     fork and join are edges of the captured graph);
 *   - errors are returned (0 = ok), never thrown (the reference's `throw "literal"` terminates the process through the
 *     Cython shim); deodr_hip_last_error() gives the message;
 *   - the adjoint never mutates `image` / `image_b` / `err_buffer` / `err_buffer_b` (the reference un-antialiases
 *     `image` in place and scales `image_b`, H.h:1738-1742); gradients are ACCUMULATED into the *_b arrays exactly as
 *     the reference does (H.h:2982, 3048, 3073, 3128);
 *   - `n_views` independent views of the same mesh (same faces / uv / texture, per-view ij / depths / colors / shade /
 *     edgeflags) are rasterized by one call: the natural batch axis of the path (deodr/mesh_fitter.py:536-546).
 *
 * Floating-point layout: "vertex" arrays (depths, uv, ij, shade, colors and their adjoints) share `vertex_dtype`;
 * "pixel" arrays (image, z_buffer, image_b, obs, err_buffer(_b), texture(_b), background_*) share `pixel_dtype`.
 * All arithmetic is done in double precision whatever the storage type.
 */
#ifndef DEODR_HIP_H
#define DEODR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEODR_HIP_F32 0
#define DEODR_HIP_F64 1

#define DEODR_HIP_MAX_COLORS 64

typedef struct DeodrHipScene
{
	/* topology, shared by all views */
	const uint32_t *faces;	  /* [T,3] */
	const uint32_t *faces_uv; /* [T,3] */
	const uint8_t *textured;  /* [T] */
	const uint8_t *shaded;	  /* [T] */
	/* per-view vertex attributes, dense [n_views, ...], dtype = vertex_dtype */
	const void *depths;		  /* [n_views, V] */
	const void *ij;			  /* [n_views, V, 2]  ij[:,0] = x (column), ij[:,1] = y (row) */
	const void *shade;		  /* [n_views, V] */
	const void *colors;		  /* [n_views, V, C] */
	const uint8_t *edgeflags; /* [n_views, T, 3] */
	const void *uv;			  /* [Vuv, 2] shared */
	/* pixel-typed inputs */
	const void *texture;		  /* [Ht, Wt, C] or NULL when no triangle is textured */
	const void *background_image; /* [n_views, H, W, C] or NULL */
	const void *background_color; /* [C] (device) or NULL; exactly one of the two backgrounds is given */
	/* adjoints (only read by deodr_hip_render_scene_b; accumulated into) */
	void *uv_b;		 /* [Vuv, 2]          vertex_dtype, summed over views */
	void *ij_b;		 /* [n_views, V, 2]   vertex_dtype */
	void *shade_b;	 /* [n_views, V]      vertex_dtype */
	void *colors_b;	 /* [n_views, V, C]   vertex_dtype */
	void *texture_b; /* [Ht, Wt, C]       pixel_dtype, summed over views; may be NULL when texture is NULL */
	int nb_triangles, nb_vertices, nb_uv;
	int height, width, nb_colors;
	int texture_height, texture_width;
	int clockwise, backface_culling, strict_edge, perspective_correct, integer_pixel_centers;
	int n_views;
	int vertex_dtype; /* DEODR_HIP_F32 / DEODR_HIP_F64 */
	int pixel_dtype;  /* DEODR_HIP_F32 / DEODR_HIP_F64 */
	int deterministic; /* non-zero: the calls on THIS scene accumulate their gradients in integers, bit-identical from run to run (see
						* deodr_hip_set_deterministic, the process-wide switch for every scene); a property of the call, no global involved */
} DeodrHipScene;

/* Size in bytes of the device workspace for a scene of these dimensions.  `pool_pairs` bounds the number of
 * (tile, primitive) pairs that spill out of the fixed per-tile lists (0 = default heuristic).  The workspace carries
 * the forward state (per-primitive records, tile lists, the per-pixel owner buffer) that the adjoint reuses; it must be
 * zero-filled once after allocation (hipMemset) and then belongs to one scene stream of calls. */
size_t deodr_hip_workspace_bytes(int nb_triangles, int height, int width, int nb_colors, int n_views, size_t pool_pairs);

/* renderScene.  image [n_views,H,W,C], z_buffer [n_views,H,W]; with antialiase_error: obs [n_views,H,W,C] (read) and
 * err_buffer [n_views,H,W] (written).  image / z_buffer may be NULL to compute only the forward state. */
int deodr_hip_render_scene(const DeodrHipScene *scene, void *image, void *z_buffer, double sigma, int antialiase_error,
						   const void *obs, void *err_buffer, void *workspace, size_t workspace_bytes, void *stream);

/* renderScene_B.  Requires backface_culling and !perspective_correct like the reference (H.h:2922, 810).
 * Extension ("residual mode", antialiase_error == 0 only): with image_b == NULL and image, obs given, the adjoint of the
 * sum-of-squares loss L = sum (image - obs)^2 is propagated, i.e. image_b = 2 (image - obs) is formed inside the kernel
 * from the rendered image (what Scene2D.render_compare_and_backward does on the host, dr.py:728-732) instead of being
 * written to and read back from HBM.
 * antialiase_error != 0 (`image` = the frame deodr_hip_render_scene wrote in this mode, `obs`, `err_buffer_b`): nb_colors <= 4 runs on
 * the LDS-staged kernels since round 6 (tiles without silhouette edges: image_b = -2 (obs - image) err_buffer_b, H.h:3054-3060; tiles with:
 * a staged sweep over the error buffer, H.h:2200-2368, 2481-2618); more channels, and the deterministic mode, on the un-staged ones.
 * have_forward_state != 0: the workspace still holds the state of the matching deodr_hip_render_scene call (same scene
 * arrays, same sigma) and is reused; 0: the forward state is recomputed first (stateless use, as the reference).  After a
 * deodr_hip_render_scene_fit on the same workspace the state is always recomputed (the fused forward does not keep the
 * owner ids of the tiles it has already back-propagated through). */
int deodr_hip_render_scene_b(const DeodrHipScene *scene, const void *image, const void *z_buffer, const void *image_b,
							 double sigma, int antialiase_error, const void *obs, const void *err_buffer,
							 const void *err_buffer_b, void *workspace, size_t workspace_bytes, int have_forward_state,
							 void *stream);

/* One step of a fit: renderScene followed by renderScene_B for the sum-of-squares loss L = sum (image - obs)^2, i.e. what
 * Scene2D.render_compare_and_backward (dr.py:700-740, antialiase_error = False branch) does with two calls and a host
 * subtraction in between.  Outputs are exactly those of deodr_hip_render_scene (image, z_buffer) followed by
 * deodr_hip_render_scene_b in residual mode (the scene's *_b arrays are accumulated into).  Because dL/dimage of a pixel
 * is known as soon as the pixel is resolved, the forward raster back-propagates through the tiles that have no silhouette
 * edge in the same pass; only the tiles with edges are visited again.  Same preconditions as deodr_hip_render_scene_b.
 * clear_gradients != 0: the scene's *_b arrays are zeroed first (Scene2D.clear_gradients, dr.py) inside the same launches,
 * so that a fit loop needs no separate fills; 0: they are accumulated into, as renderScene_B does.
 * STREAMS.  Everything is queued on `stream`, with two exceptions that go through ONE library-owned side stream per device (forked from and
 * joined back to `stream` by events, so that the call still looks ordered on `stream` to the caller): the background fill of a forward-only
 * deodr_hip_render_scene, and -- a TEXTURED fit step of 8 views or more -- the head walkers of the forward raster (a kernel of their own at
 * three waves per SIMD; everybody else runs beside it at four).  Callers that drive ONE device from several host threads or streams are
 * therefore serialised through that side stream (and a mutex held across its two launches) for those calls; results are unaffected.  Under
 * stream capture the textured fit step takes its ONE-kernel form (a captured graph has no second stream to fork to), so eager and
 * captured runs of such a step launch different kernel instances: equal results up to the order of the float atomics, different timings
 * -- compare eager with eager, replay with replay. */
int deodr_hip_render_scene_fit(const DeodrHipScene *scene, void *image, void *z_buffer, double sigma, const void *obs,
							   int clear_gradients, void *workspace, size_t workspace_bytes, void *stream);

/* The same step with options (NULL: none).
 *
 * The loss -- `err` of Scene2D.render_compare_and_backward (dr.py:725-734): loss[0] = sum over views, pixels and channels of
 * (image - obs)^2, of the frame as stored (rounded to the pixel type) -- WITHOUT a pass over the frame: the tile walkers of the
 * forward raster have every residual in registers.  What they cannot see, the background of the tiles that hold no primitive, is
 * accounted for by a table computed ONCE per (observation, background, clamp) by deodr_hip_background_loss: tile_loss[0] = the loss
 * of a frame batch that is all background, tile_loss[1 + view * ntiles + t] = that of the 8 x 8-pixel tile t of a view
 * (deodr_hip_fit_loss_bytes(H, W, n_views) bytes); the step returns tile_loss[0] + sum over the non-empty tiles of (loss of the tile
 * - tile_loss[tile]).  loss_scratch: device memory of deodr_hip_fit_loss_bytes bytes (no initialisation needed).  Scenes the staged
 * kernels do not take (more than 4 channels) or without triangles get the loss from one pass over the finished frame instead.
 *
 * clamp: the residual is that of L = sum (clamp(image, clamp_lo, clamp_hi) - obs)^2 -- the data term of the reference's depth fitter
 * (deodr/mesh_fitter.py:108-123: the rendered depth image is clipped to [0, max_depth] before it is compared) --, its gradient
 * passing on the closed interval as NumPy's / torch's clip does.  The stored image is the un-clamped rendering. */
typedef struct DeodrHipFitOptions
{
	const double *tile_loss; /* table of deodr_hip_background_loss (made with the same clamp), or NULL: no loss wanted */
	double *loss;			 /* [1], device */
	void *loss_scratch;		 /* deodr_hip_fit_loss_bytes bytes, device */
	int clamp;
	double clamp_lo, clamp_hi;
	/* Step-done flag, or NULL (a 4-byte word of device memory the caller owns): when the gradients of this step are complete and visible
	 * device-wide, the step's last wavefront stores done_value there.  A consumer on ANOTHER stream -- the shared-gradient reduction of a
	 * sharded fit -- waits for it with deodr_hip_wait_flag instead of a hipEvent: an event recorded on the render stream and waited for by
	 * a second queue costs the render stream ~8 us per step on MI355X (tools/dist_overhead_probe.py), this flag nothing.  Use increasing
	 * values (the step number): deodr_hip_wait_flag waits for *flag >= value in serial-number arithmetic.
	 * WHAT THE FLAG COVERS: the gradient arrays (ij_b, colors_b, shade_b, uv_b, texture_b) -- they are written by memory-side atomics, which are
	 * visible device-wide once acknowledged, and the flag is stored behind their acknowledgement.  NOT covered: *loss, the image and the
	 * z-buffer of the step (plain stores, which may still sit in an XCD's L2 when the flag is seen): a consumer that reads those orders
	 * itself behind the step's stream (an event, or stream order), as before. */
	uint32_t *done_flag;
	uint32_t done_value;
} DeodrHipFitOptions;
size_t deodr_hip_fit_loss_bytes(int height, int width, int n_views);
int deodr_hip_background_loss(const DeodrHipScene *scene, const void *obs, const DeodrHipFitOptions *options, double *tile_loss, void *workspace,
							  size_t workspace_bytes, void *stream);
int deodr_hip_render_scene_fit_ex(const DeodrHipScene *scene, void *image, void *z_buffer, double sigma, const void *obs, int clear_gradients,
								  const DeodrHipFitOptions *options, void *workspace, size_t workspace_bytes, void *stream);

/* ---- Front half of a fit iteration: the O(V) algebra between the parameters of a fitter and the 2.5-D
 * scene, and its adjoint, as kernels -- as torch ops one iteration is ~240 launches, most of them this algebra.  Plain double
 * arrays on the device, contiguous, asynchronous on `stream`; n = number of views (poses / cameras), V vertices, T triangles.
 *
 * deodr_hip_rigid_transform     out[b][v] = qrot(quaternions[b], vertices[v]) + translations[b]   (q = (x, y, z, w), unit;
 *                               deodr/tools.py:8-22, deodr/mesh_fitter.py:139-151)
 * deodr_hip_rigid_transform_b   its adjoint (deodr/tools.py:25-35): vertices_b [V,3] = sum over the views; pose_b [n*4 + n*3] = the
 *                               quaternion adjoints of all views, then the translation adjoints (both overwritten)
 * deodr_hip_project_points      Camera.project_points (deodr/differentiable_renderer.py:341-395): points [n,V,3], extrinsic [n,3,4],
 *                               intrinsic [n,3,3], distortion [n,5] (k1, k2, p1, p2, k3) or NULL -> ij [n,V,2] (x = column first),
 *                               depths [n,V]
 * deodr_hip_project_points_b    Camera.project_points_backward (dr.py:397-438): ij_b, depths_b (or NULL) -> points_b [n,V,3]
 * deodr_hip_silhouette_flags    TriMeshAdjacencies.edge_on_silhouette (deodr/triangulated_mesh.py:153-166) for n views: flags
 *                               [n,T,3] = 1 where exactly one of the faces on edge e of face f is front-facing in the image;
 *                               edge_faces [T,3] = the face across edge (v_e, v_e+1) of face f, 0xffffffff on a boundary
 * deodr_hip_momentum_update     s = (1 - damping)(inertia s + (1 - inertia) clamp(-factor (grad + grad2), +-step_max)); x += s
 *                               (deodr/mesh_fitter.py:153-190) for up to 8 parameter tensors in one launch; step_max <= 0: no clamp;
 *                               normalize_rows[k] = r > 0: x[k] is [count/r, r] and every row is renormalised afterwards (the
 *                               quaternions, mesh_fitter.py:176; per view: see DESIGN.md section 6, divergence 6) */
int deodr_hip_rigid_transform(const double *vertices, const double *quaternions, const double *translations, double *out, int V, int n, void *stream);
int deodr_hip_rigid_transform_b(const double *vertices, const double *quaternions, const double *out_b, double *vertices_b, double *pose_b, int V, int n,
								void *stream);
int deodr_hip_project_points(const double *points, const double *extrinsic, const double *intrinsic, const double *distortion, double *ij, double *depths,
							 int V, int n, void *stream);
int deodr_hip_project_points_b(const double *points, const double *extrinsic, const double *intrinsic, const double *distortion, const double *ij_b,
							   const double *depths_b, double *points_b, int V, int n, void *stream);
int deodr_hip_silhouette_flags(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T, int V, int n, int clockwise,
							   void *stream);
int deodr_hip_momentum_update(int n_tensors, double *const *x, double *const *speed, const double *const *grad, const double *const *grad2,
							  const double *factor, const double *step_max, const int *count, const int *normalize_rows, double inertia, double damping,
							  const double *grad_scale, const double *const *grad_mean, double *const *mean_out, double *energy, const double *data_energy,
							  double data_weight, void *scratch, size_t scratch_bytes, void *stream);

/* ---- One fit iteration without an autograd graph (deodr/mesh_fitter.py:108-190, 287-376, 529-632): the chain parameters -> posed and
 * projected vertices -> shading -> [silhouette flags, deodr_hip_render_scene_fit] -> adjoints -> rigid energy -> momentum update, about
 * twelve launches.  Every sum over vertices is deterministic (per-workgroup partials added up in a fixed order by the last workgroup
 * to arrive), every gather runs over a static vertex -> (face, corner) table: no atomics on values.  `scratch`: device memory of
 * deodr_hip_fit_scratch_bytes(V, n) bytes, ZERO-FILLED ONCE by the caller (the kernels leave its counter words zero), shared by all
 * these calls on one stream.
 *
 * deodr_hip_fit_pose_project     vertices [V,3] (centred IN PLACE by vertices_mean [3] when not NULL, mesh_fitter.py:131), raw
 *                                quaternions [n,4] (normalised inside), translations [n,3], cameras as in deodr_hip_project_points
 *                                -> posed [n,V,3], ij [n,V,2], depths [n,V]; depth_colors [n,V] (or NULL) = depth_scale * depths, the
 *                                one-channel "colour" a depth image is rendered from (Scene3D.render_depth, dr.py:1001-1036)
 * deodr_hip_fit_pose_project_b   posed_b [n,V,3] (or NULL), ij_b, depths_b_scale * depths_b (or NULL) -> vertices_b [V,3] summed over the views;
 *                                out [3 + 7 n] = column mean of vertices_b (the data gradient is projected on zero-mean displacements,
 *                                mesh_fitter.py:140, 319), quaternion adjoints [n,4] w.r.t. the RAW quaternions, translation adjoints [n,3];
 *                                colors_sum [V,C] (or NULL) = colors_b [n,V,C] summed over the views (per-vertex colours shared by the
 *                                views: mesh_fitter.py:518-527 adds them up view by view on the host)
 * deodr_hip_vertex_shade         posed [n,V,3] -> luminosity [n,V] = max(0, -normal . light) + ambient (dr.py:814-822) with the vertex
 *                                normals of triangulated_mesh.py:113-151, and/or colors [n,V,C] = color [C] * luminosity (C <= 3).
 *                                vf_offsets [V+1], vf_corners [3T]: for every vertex the slots 3 f + corner it occupies in `faces`
 * deodr_hip_vertex_shade_b       luminosity_b and/or colors_b -> posed_b [n,V,3] (overwritten); out [4 + C] = light_b [3], ambient_b,
 *                                color_b [C]
 * deodr_hip_rigid_energy         energy[0] = 0.5 c d^T (L^T L) d, d = vertices - vertices_ref, and its gradient c (L^T L) d [V,3]
 *                                (deodr/laplacian_rigid_energy.py:15-41); L^T L as CSR rows m_offsets [V+1], m_cols, m_vals.  With
 *                                data_energy [1] != NULL also energy[1] = data_weight * data_energy[0] + energy[0], the energy a
 *                                fitter's step reports (mesh_fitter.py:147)
 * deodr_hip_depth_residual       the data term of the depth fitter (mesh_fitter.py:108-123) over `count` pixels of a rendered depth image in
 *                                the pixel type: depth = clamp(image, 0, max_depth), diff = (depth - obs)^2, loss[0] = sum diff, image_b =
 *                                2 (depth - obs) where 0 <= image <= max_depth, else 0 (pixel type) -- what deodr_hip_render_scene_b takes
 * deodr_hip_l2_loss              out[0] = sum (image - obs)^2 over `count` values of the pixel type (DEODR_HIP_F32 / _F64), accumulated in
 *                                double: the data energy whose gradient deodr_hip_render_scene_fit back-propagates (mesh_fitter.py:296-318)
 * deodr_hip_momentum_update      (above) grad_scale[k]: weight of grad (not of grad2); grad_mean[k] [3] or NULL: subtracted from every
 *                                row of a [count/3, 3] gradient; mean_out[k] [3] or NULL: column mean of the updated tensor; energy [2] or NULL:
 *                                energy[1] = data_weight * data_energy[0] + energy[0] (mesh_fitter.py:147; energy[0] from deodr_hip_fit_front)
 * deodr_hip_fit_front            what lies between deodr_hip_fit_pose_project and the rasterizer, in ONE launch (none of the three depends on
 *                                another): deodr_hip_silhouette_flags (flags != NULL), deodr_hip_vertex_shade (luminosity or colors != NULL) and
 *                                the rigid energy with its gradient (gradient != NULL; energy[0] written, energy[1] left to
 *                                deodr_hip_momentum_update: the data energy is not known yet).  Same results, bit for bit, as the three calls */
size_t deodr_hip_fit_scratch_bytes(int V, int n);
int deodr_hip_fit_front(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T, const double *posed,
						const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light, const double *ambient, const double *color, int C,
						double *luminosity, double *colors, const double *vertices, const double *vertices_ref, const uint32_t *m_offsets, const uint32_t *m_cols,
						const double *m_vals, double cregu, double *gradient, double *energy, void *scratch, size_t scratch_bytes, int V, int n, int clockwise,
						void *stream);
int deodr_hip_fit_pose_project(double *vertices, const double *vertices_mean, const double *quaternions, const double *translations, const double *extrinsic,
							   const double *intrinsic, const double *distortion, double *posed, double *ij, double *depths, double *depth_colors,
							   double depth_scale, int V, int n, void *stream);
int deodr_hip_fit_pose_project_b(const double *vertices, const double *quaternions, const double *posed, const double *extrinsic, const double *intrinsic,
								 const double *distortion, const double *posed_b, const double *ij_b, const double *depths_b, double depths_b_scale,
								 double *vertices_b, double *out, void *scratch, size_t scratch_bytes, int V, int n, const double *colors_b, int nb_colors,
								 double *colors_sum, void *stream);
int deodr_hip_vertex_shade(const double *posed, const uint32_t *faces, const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light,
						   const double *ambient, const double *color, int C, double *luminosity, double *colors, int V, int n, int clockwise, void *stream);
int deodr_hip_vertex_shade_b(const double *posed, const uint32_t *faces, const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light,
							 const double *ambient, const double *color, int C, const double *luminosity_b, const double *colors_b, double *posed_b, double *out,
							 void *scratch, size_t scratch_bytes, int V, int n, int clockwise, void *stream);
int deodr_hip_rigid_energy(const double *vertices, const double *vertices_ref, const uint32_t *m_offsets, const uint32_t *m_cols, const double *m_vals,
						   double cregu, double *gradient, double *energy, const double *data_energy, double data_weight, void *scratch, size_t scratch_bytes,
						   int V, void *stream);
int deodr_hip_l2_loss(const void *image, const void *obs, int pixel_dtype, size_t count, double *out, void *scratch, size_t scratch_bytes, void *stream);
int deodr_hip_depth_residual(const void *image, int pixel_dtype, const double *obs, double max_depth, size_t count, double *depth, double *diff, void *image_b,
							 double *loss, void *scratch, size_t scratch_bytes, void *stream);

/* Bits of the sticky scene-error word: the index checks of the reference's checkSceneValid
 * (DifferentiableRenderer.h:2700-2712: `faces` entries < nb_vertices, `faces_uv` entries < nb_uv; plus the null-texture
 * check of H.h:2687-2694 that needs per-triangle data) are made by the set-up kernel where it reads the indices -- a
 * device-resident scene is never copied to the host to be validated.  An offending triangle is dropped (never
 * dereferenced) and the word is raised; the reference throws instead. */
#define DEODR_HIP_ERR_FACES 1	   /* an entry of faces is >= nb_vertices */
#define DEODR_HIP_ERR_FACES_UV 2   /* an entry of faces_uv is >= nb_uv */
#define DEODR_HIP_ERR_NO_TEXTURE 4 /* textured[k] && shaded[k] for some k although scene.texture == NULL */
#define DEODR_HIP_ERR_INTERNAL 8   /* reserved: raised by the experimental build that finalizes under the forward raster (tools/variants/finalize_in_forward.patch) when a
                                      finalize workgroup gives up waiting for the tile walkers; the product never sets it */
#define DEODR_HIP_ERR_DET_RANGE 16 /* deterministic mode (deodr_hip_set_deterministic): a contribution or a running sum left the fixed-point range
                                      +- 2^31 (or was NaN) -- the gradients of that call are wrong; not a property of the scene's indices */

/* Synchronises `stream` and reports (1) whether any forward since the workspace was zero-filled overflowed the spill pool
 * (then that result was incomplete and the call must be repeated with a workspace sized for a larger `pool_pairs`):
 * *needed_pairs receives the largest number of spilled pairs seen in any view; (2) the union of the DEODR_HIP_ERR_* bits
 * raised by any forward (*scene_errors; 0 = the scene passed checkSceneValid's index checks).  Costs a device
 * synchronisation: call it after the first render of a scene, or poll instead: the first 64 bytes of the workspace are a
 * status block of sixteen uint32 whose words [11] and [12] hold the same two values for all views (max of needed pairs, union
 * of error bits) after every forward -- an asynchronous 64-byte copy to pinned memory, inspected later, costs no
 * synchronisation (deodr_amd.hip_renderer.HipRasterizer does that).  deodr_hip_workspace_pool_pairs gives the capacity the
 * first value has to be compared with. */
int deodr_hip_workspace_status(const DeodrHipScene *scene, void *workspace, size_t workspace_bytes, void *stream, int *overflowed,
							   unsigned long long *needed_pairs, int *scene_errors);
int deodr_hip_workspace_pool_pairs(const DeodrHipScene *scene, size_t workspace_bytes, unsigned long long *pool_pairs);
#define DEODR_HIP_STATUS_WORD_NEEDED_PAIRS 11
#define DEODR_HIP_STATUS_WORD_SCENE_ERRORS 12

/* Measurement hooks (bench.py): deodr_hip_profile_enable(n), n > 0: the kernel launches of every n-th forward (and of the
 * adjoint that follows it) are bracketed by hipEvents recorded on the launch stream; 0: off.  An event pair costs ~3 us of
 * stream time on this part, i.e. ~10 % of a four-kernel step if every launch is timed, hence the sampling.
 * deodr_hip_profile_read waits for the events and returns, per kernel
 *   [0] setup_bin_kernel  [1] raster_fwd_kernel  [2] raster_bwd_kernel (all adjoint raster kernels)  [3] finalize_kernel
 * the summed elapsed milliseconds and the number of timed launches since the previous read.  Not thread-safe. */
int deodr_hip_profile_enable(int every);
int deodr_hip_profile_read(double ms_sum[4], unsigned long long launches[4]);
/* The same measurement without event packets between the launches (a hipEvent pair per kernel costs the step it measures ~ 36 us: with the
 * launches of every 4th step bracketed, a 20-step run reads 7 % slow).  device_buffer: rows x 4 uint64 on the device, zero-filled by the
 * caller; forward number r (counted from this call) writes row r: [0] the 100 MHz realtime counter (10 ns ticks) when the first thread of
 * setup_bin_kernel starts, [1] tile_scan_kernel, [2] finalize_kernel (0 where a kernel did not run).  Kernels of one stream run back to
 * back, so set-up lasts [1] - [0], tile scan + forward raster (+ the adjoint raster kernels of a two-call step) [2] - [1], finalize until
 * the [0] of the next step.  Rows beyond `rows` are not written; NULL / 0 switches it off.  Not thread-safe. */
int deodr_hip_profile_stamps(void *device_buffer, int rows);

/* Measurement hook (bench.py's "necessary bytes"): synchronises `stream` and counts, over all views of the last forward, the
 * 8 x 8-pixel tiles that received at least one primitive and those that hold silhouette edges. */
int deodr_hip_workspace_census(const DeodrHipScene *scene, void *workspace, size_t workspace_bytes, void *stream,
							   unsigned long long *nonempty_tiles, unsigned long long *edge_tiles);

/* Measurement hook (bench.py's hbm_probe): `reps` back-to-back device-to-device copies of `bytes` bytes (a multiple of 16, both
 * pointers 16-byte aligned) with the library's own streaming 16-byte non-temporal loads and stores -- the ceiling the fill and frame
 * stores of the rasterizer live under on THIS box, next to the 8 TB/s of the data sheet.  mode 0: copy, 1: write only (src unused),
 * 2: read only (dst: 8 bytes receiving a checksum so that the loads are not removed).  Asynchronous on `stream`. */
int deodr_hip_copy_probe(void *dst, const void *src, size_t bytes, int mode, int reps, void *stream);

/* Test hook: non-zero makes every call use the generic (un-staged) kernels that otherwise only serve nb_colors > 4 and
 * antialiase_error, so that the parity suite can exercise both code paths on the same scenes.  This (and the profiling
 * hook above) is the only process-wide state; the library reads NO environment variable. */
int deodr_hip_force_generic(int on);

/* Deterministic accumulation (the reference is bit-reproducible by construction -- one thread,
 * DifferentiableRenderer.h:1029-1037).  Non-zero: every later call runs the un-staged kernels with INTEGER accumulation -- each
 * contribution to a moment accumulator, a vertex gradient or the texture gradient is rounded to a multiple of 2^-32 and added as a
 * 64-bit integer, so the order in which the memory system executes the atomics no longer shows: gradients (and, in a fit step, the
 * loss) are bit-identical from run to run.  Limits: |any gradient sum| < 2^31 (a contribution or a running sum beyond it raises the
 * sticky bit DEODR_HIP_ERR_DET_RANGE of the status block instead of wrapping silently), resolution 2^-32 (~2.3e-10) per contribution;
 * several times slower than the default path (it is a mode for tests and for debugging an optimiser, default off); the library
 * allocates an int64 shadow of the gradient arrays the first time a (device, stream) pair is used in this mode (hipMalloc: not under
 * stream capture; calls on different streams or host threads never share a shadow).  DeodrHipScene::deterministic asks for the mode per
 * scene (re-entrant: nothing but the arguments of the call decides); this switch turns it on for every scene of the process, like
 * deodr_hip_force_generic. */
int deodr_hip_set_deterministic(int on);

/* A sharded multi-view fit (deodr/mesh_fitter.py:504-546 with the frames dealt to ranks): what the ranks all-reduce is the gradient of what
 * the views share -- the mesh vertices and their colours.
 * deodr_hip_views_gradient_sum   the adjoint of every view's camera projection (deodr_hip_project_points_b's formulas) applied to ij_b [n,V,2]
 *                                (and depths_b_scale * depths_b [n,V], or NULL) and summed over the n views -> vertices_b [V,3]; colors_b
 *                                [n,V,C] summed over the views -> colors_sum [V,C] (or NULL): `vertices_b += ...` of mesh_fitter.py:518-527 in
 *                                one launch, straight into the packed buffer the collective runs on.
 * deodr_hip_wait_flag            queues, on `stream`, a one-wavefront kernel that returns when *flag >= value (DeodrHipFitOptions::done_flag; the
 *                                producer must have been queued BEFORE this call, on any stream of the device: the wait then never depends on
 *                                work that does not exist yet).  After timeout_seconds it gives up and sets status[0] = 1 (device memory, or
 *                                NULL); later waits on a status word that is set return at once -- the caller checks it where it synchronises. */
int deodr_hip_views_gradient_sum(const double *posed, const double *extrinsic, const double *intrinsic, const double *distortion, const double *ij_b,
								 const double *depths_b, double depths_b_scale, double *vertices_b, int V, int n, const double *colors_b, int nb_colors,
								 double *colors_sum, void *stream);
int deodr_hip_wait_flag(const uint32_t *flag, uint32_t value, uint32_t *status, double timeout_seconds, void *stream);

/* Message of the last error returned on this host thread. */
const char *deodr_hip_last_error(void);

/* ABI version of this header; bumped on any incompatible change. */
int deodr_hip_abi_version(void);
#define DEODR_HIP_ABI_VERSION 12

#ifdef __cplusplus
}
#endif
#endif /* DEODR_HIP_H */
