"""Box drawing for the reference's viewer helpers (detection/plot/draw_box_label.py, get_color.py) -- numpy only (no OpenCV /
matplotlib in this image): rectangle outlines, no text rendering.  Host-side convenience, not on the hot path."""
import numpy as np

__all__ = ['draw_box_label', 'get_color']

_PALETTE = [(255, 56, 56), (255, 157, 151), (255, 112, 31), (255, 178, 29), (207, 210, 49), (72, 249, 10), (146, 204, 23),
            (61, 219, 134), (26, 147, 52), (0, 212, 187), (44, 153, 168), (0, 194, 255), (52, 69, 147), (100, 115, 255),
            (0, 24, 236), (132, 56, 255), (82, 0, 133), (203, 56, 255), (255, 149, 200), (255, 55, 199)]


def get_color(idx, bgr=True):
    c = _PALETTE[int(idx) % len(_PALETTE)]
    return (c[2], c[1], c[0]) if bgr else c


def draw_box_label(image, box, text='', line_width=2, line_color=(128, 128, 128), font_size=1, font_color=(255, 255, 255), bgr=True):
    """Draw the xyxy ``box`` into the HxWx3 uint8 ``image`` in place (and return it).  ``text`` is accepted for signature
    compatibility; glyph rendering needs a font rasteriser this image does not have."""
    assert isinstance(image, np.ndarray), f'Type of parameter image must be np.ndarray, not {type(image)}'
    if isinstance(line_color, int):
        line_color = get_color(line_color, bgr=bgr)
    h, w = image.shape[:2]
    x0, y0, x1, y1 = (int(round(float(v))) for v in box)
    x0, x1 = max(0, min(x0, w - 1)), max(0, min(x1, w - 1))
    y0, y1 = max(0, min(y0, h - 1)), max(0, min(y1, h - 1))
    t = max(1, int(line_width))
    image[y0:y0 + t, x0:x1 + 1] = line_color
    image[max(y1 - t + 1, 0):y1 + 1, x0:x1 + 1] = line_color
    image[y0:y1 + 1, x0:x0 + t] = line_color
    image[y0:y1 + 1, max(x1 - t + 1, 0):x1 + 1] = line_color
    return image
