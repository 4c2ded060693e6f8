"""Keep the interactive viser viewer alive without making the browser rasterize splats
(SURVEY.md 8f rank 4).

The reference registers every splat group with ``server.scene.add_gaussian_splats``
(sim_a_splat/splat/splat_handler.py:106-143) and lets each browser sort and draw them in WebGL.
Here the groups live in a ``SplatScene`` on the GPU; the bridge renders the view of each
connected client with the HIP rasterizer and pushes the frame as that client's background image.
Meshes, frames and GUI the caller adds to the same server keep working unchanged.

viser is imported by nobody here: the bridge only needs the handful of attributes below, so it
runs against ``viser.ViserServer`` (0.2.x) and against the stand-ins used by the tests.

    server.on_client_connect(cb) / server.on_client_disconnect(cb)     cb(client)
    client.client_id
    client.camera.wxyz, .position, .fov, .aspect                       c2w, OpenCV axes (row T7)
    client.camera.on_update(cb)                                        cb(camera)
    client.scene.set_background_image(image, format=..., jpeg_quality=...)
"""
from __future__ import annotations

import threading
from typing import Dict, Optional

import numpy as np

from .scene import SplatScene


class ViserBridge:
    def __init__(self, server, scene: SplatScene, height: int = 720, max_width: int = 2560, image_format: str = "jpeg",
                 jpeg_quality: int = 85):
        self.server, self.scene = server, scene
        self.height, self.max_width = int(height), int(max_width)
        self.image_format, self.jpeg_quality = image_format, int(jpeg_quality)
        self._clients: Dict[int, object] = {}
        self._lock = threading.Lock()           # one rasterizer context, one caller at a time (8b threading)
        self.frames_pushed = 0
        server.on_client_connect(self._connect)
        if hasattr(server, "on_client_disconnect"):
            server.on_client_disconnect(self._disconnect)

    # -- client life cycle -----------------------------------------------------------------------
    def _connect(self, client) -> None:
        self._clients[client.client_id] = client
        client.camera.on_update(lambda _cam, c=client: self.push(c))
        self.push(client)

    def _disconnect(self, client) -> None:
        self._clients.pop(client.client_id, None)

    # -- rendering -------------------------------------------------------------------------------
    def frame_for(self, client) -> np.ndarray:
        cam = client.camera
        aspect = float(getattr(cam, "aspect", 16.0 / 9.0) or 16.0 / 9.0)
        width = int(min(self.max_width, max(16, round(self.height * aspect))))
        with self._lock:
            return self.scene.get_render(self.height, width, np.asarray(cam.wxyz, dtype=np.float64),
                                         np.asarray(cam.position, dtype=np.float64), fov=float(cam.fov))

    def push(self, client) -> Optional[np.ndarray]:
        """Render the client's current view and set it as its background image."""
        frame = self.frame_for(client)
        client.scene.set_background_image(frame, format=self.image_format, jpeg_quality=self.jpeg_quality)
        self.frames_pushed += 1
        return frame

    def refresh(self) -> int:
        """Re-render every connected client (call after group poses changed, e.g. from ``draw_handler``)."""
        clients = list(self._clients.values())
        for c in clients:
            self.push(c)
        return len(clients)
